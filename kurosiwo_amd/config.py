"""Config / flag system of the reference on the same file names, keys and derived keys
(/root/reference/main.py:29-76, utilities/utilities.py:15-70 and :350-422).

New optional keys understood by this build: "precision" ("bf16" | "fp32"), "synthetic_tiles".
"""
import json
import re
from datetime import datetime
from pathlib import Path


def load_json5(path):
    """The reference's config files are JSON5 (// comments, trailing commas, pyjson5); this loader
    accepts both those and strict JSON."""
    txt = open(path, "r").read()
    txt = re.sub(r'("(?:\\.|[^"\\])*")|//[^\n]*', lambda m: m.group(1) or "", txt)
    txt = re.sub(r",(\s*[}\]])", r"\1", txt)
    return json.loads(txt)


CLASS_WEIGHTS_RANDOM_EVENTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]   # utilities.py:393-397


def update_config(config, args=None, root="."):
    """Merge data/train configs and derive num_channels, class_weights, device (utilities.py:350-412)."""
    root = Path(root)
    config.update(load_json5(root / "configs/train/data_config.json"))
    if args is not None:
        if getattr(args, "inputs", None) is not None:
            config["inputs"] = args.inputs
        if getattr(args, "dem", False):
            config["dem"] = args.dem
            if getattr(args, "slope", False):
                config["slope"] = args.slope
    config.update(load_json5(root / "configs/train/train_config.json"))
    if config["task"] == "cd" or config["method"] == "convlstm":
        config["num_channels"] = len(config["channels"])
    else:
        config["num_channels"] = len(config["channels"]) * len(config["inputs"])
    if config["dem"]:
        config["num_channels"] += 1
    if config.get("slc"):
        config["num_channels"] = (config["num_channels"] - 1) * 2 + 1 if config["dem"] else config["num_channels"] * 2
    if config["weighted"] and config["track"] == "RandomEvents":
        config["class_weights"] = list(CLASS_WEIGHTS_RANDOM_EVENTS)
    else:
        config["class_weights"] = [1.0, 1.0, 1.0]
    config["device"] = f'cuda:{config["gpu"]}' if config["gpu"] is not None else "cpu"
    print("train activations ", len(config["train_acts"]))
    print("val activations ", len(config["val_acts"]))
    print("test activations ", len(config["test_acts"]))
    print("Configs updated")
    return config


def create_checkpoint_directory(configs, model_configs):
    """Same directory naming as utilities.py:15-70 for the tasks in scope."""
    if configs["task"] == "cd":
        ts = datetime.now().strftime("%Y%m%d%H%M%S")
        path = f'checkpoints/{configs["method"].lower()}/{configs["track"]}_{ts}'
    elif configs["task"] == "segmentation":
        if model_configs.get("backbone"):
            path = (f'checkpoints/{model_configs["architecture"]}/{model_configs["backbone"]}/'
                    f'{"-".join(configs["channels"])}_patches_{len(configs["inputs"])}/{configs["track"]}')
        else:
            path = f'checkpoints/{model_configs["architecture"]}'
    else:
        path = f'checkpoints/{configs["task"]}'
    Path(path).mkdir(parents=True, exist_ok=True)
    return path


def init_lr_scheduler(optimizer, configs, model_configs, model_name=None, steps=None):
    """utilities.py:268-304 ('step' is broken in the reference and raises here too)."""
    import torch
    sched = model_configs[model_name]["lr_schedule"] if model_name is not None else model_configs["lr_schedule"]
    if sched == "cosine":
        s = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, steps)
    elif sched is None:
        s = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda _: 1, last_epoch=-1)
    elif sched == "linear":
        s = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda e: 1.0 - e / float(configs["epochs"] + 1))
    else:
        raise NotImplementedError(f"{sched} LR scheduling is not yet implemented!")
    if configs.get("resume_checkpoint"):
        s.load_state_dict(torch.load(configs["resume_checkpoint"])["lr_scheduler_state_dict"])
    return s
