#!/usr/bin/env python3
"""Entry point with the reference's CLI (/root/reference/main.py:29-36) and dispatch (:83-195)
for the tasks/methods that have a HIP implementation: task "cd" method "snunet", task "segmentation" method
"finetune" (FloodViT).

  python main.py --method snunet --inputs pre_event_1 post_event [--dem] [--slope] [--batch_size N] [--seed S]
  python main.py --method finetune --inputs pre_event_1 pre_event_2 post_event [--batch_size N]
  python main.py --method mae [--batch_size N]        (MAE pre-training of the FloodViT encoder)
"""
import argparse
import pprint
import random
from pathlib import Path

import numpy as np
import torch

from kurosiwo_amd.config import create_checkpoint_directory, load_json5, update_config
from kurosiwo_amd.data import prepare_loaders
from kurosiwo_amd.distributed import barrier, broadcast_object, init_distributed, is_main
from kurosiwo_amd.model_utilities import initialize_cd_model, initialize_segmentation_model
from kurosiwo_amd.training.change_detection_trainer import eval_change_detection, train_change_detection
from kurosiwo_amd.training.segmentation_trainer import eval_semantic_segmentation, train_semantic_segmentation

parser = argparse.ArgumentParser()
parser.add_argument("--method", default=None)
parser.add_argument("--backbone", default=None)
parser.add_argument("--dem", action="store_true", default=False)
parser.add_argument("--slope", action="store_true", default=False)
parser.add_argument("--batch_size", default=None)
parser.add_argument("--inputs", nargs="+", default=None)
parser.add_argument("--seed", type=int, default=999)
parser.add_argument("--hip_graph", action="store_true", default=False,
                    help="replay the fused train step as one captured HIP graph (single GPU; kurosiwo_amd/trainer.py)")


def main(argv=None):
    args = parser.parse_args(argv)
    np.random.seed(args.seed)
    random.seed(args.seed)
    torch.manual_seed(args.seed)
    configs = load_json5("configs/config.json")
    if args.method is not None:
        configs["method"] = args.method
    name = configs["method"].lower()
    model_configs = load_json5(f'configs/method/{name}/{name.replace("-", "_")}.json')
    if args.backbone is not None:
        model_configs["backbone"] = args.backbone
    configs.update(model_configs)
    configs = update_config(configs, args)          # (the reference drops --dem without --inputs: main.py:66-69 bug, not kept)
    if args.hip_graph:
        configs["hip_graph"] = True
    # data parallelism: under `python -m torch.distributed.run --nproc-per-node N main.py ...` every process joins
    # the RCCL group, uses cuda:LOCAL_RANK and trains on its contiguous shard of each global batch (kurosiwo_amd/distributed.py)
    rank, local_rank, world = init_distributed(configs)
    if world > 1 and int(args.batch_size or configs["batch_size"]) % world:
        raise SystemExit(f'batch_size {args.batch_size or configs["batch_size"]} must be divisible by the world size {world}')
    if name == "mae" or configs.get("task") == "mae":
        # main.py:160-163 of the reference: task "mae" -> training.train_mae.train(configs)
        from kurosiwo_amd.training import train_mae
        configs["task"] = "mae"
        configs["num_channels"] = len(configs["channels"])
        configs["checkpoint_path"] = create_checkpoint_directory(configs, model_configs)
        if args.batch_size is not None:
            configs["batch_size"] = int(args.batch_size)
        pprint.pprint(configs)
        train_mae.train(configs)
        return 0.0
    task = "cd" if name in ("snunet", "changeformer", "siam-conc", "siam-diff", "bit-cd", "hfa-net", "adhr-cdnet") else "segmentation"
    if configs.get("task") != task:
        # the method decides the task; re-derive num_channels for it exactly as utilities/utilities.py:377-390 does
        # (cd: one date's channels; segmentation: channel concat of the selected dates; + dem; SLC doubles the SAR channels)
        configs["task"] = task
        nch = len(configs["channels"]) * (1 if task == "cd" else len(configs["inputs"])) + (1 if configs["dem"] else 0)
        if configs.get("slc"):
            nch = (nch - 1) * 2 + 1 if configs["dem"] else nch * 2
        configs["num_channels"] = nch
    configs["checkpoint_path"] = create_checkpoint_directory(configs, model_configs) if is_main() else None
    configs["checkpoint_path"] = broadcast_object(configs["checkpoint_path"])      # (time-stamped: rank 0 names it)
    if args.batch_size is not None:
        configs["batch_size"] = int(args.batch_size)
    if is_main():
        pprint.pprint(configs)
    train_loader, val_loader, test_loader = prepare_loaders(configs)
    if configs["task"] == "cd":
        if not configs["test"]:
            model = initialize_cd_model(configs, model_configs, "train")
            train_change_detection(model, train_loader, val_loader, test_loader, configs=configs, model_configs=model_configs)
        barrier()
        model = initialize_cd_model(configs, model_configs, "test")
        ckpt_path = Path(configs["checkpoint_path"]) / "best_segmentation.pt"
        if is_main():
            print(f"Loading model from: {ckpt_path}")
        checkpoint = torch.load(ckpt_path, map_location=configs["device"])
        model.load_state_dict(checkpoint["model_state_dict"])
        test_acc, test_score, miou = eval_change_detection(model, test_loader, settype="Test", configs=configs,
                                                           model_configs=model_configs)
        if is_main():
            print(f"Test mIoU: {miou}")
        return float(miou)
    if configs["task"] == "segmentation":
        model = initialize_segmentation_model(configs, model_configs)
        if not configs["test"]:
            train_semantic_segmentation(model, train_loader, val_loader, test_loader, configs=configs, model_configs=model_configs)
        barrier()
        ckpt_path = Path(configs["checkpoint_path"]) / "best_segmentation.pt"
        if is_main():
            print(f"Loading model from: {ckpt_path}")
        model = torch.load(ckpt_path, map_location=configs["device"], weights_only=False)       # whole-module pickle (main.py:151)
        test_acc, test_score, miou = eval_semantic_segmentation(model, test_loader, settype="Test", configs=configs,
                                                                model_configs=model_configs)
        if is_main():
            print(f"Test Mean IOU: {miou}")
        return float(miou)
    raise SystemExit(f'task {configs["task"]!r} is not implemented by this build (SURVEY.md §8)')


if __name__ == "__main__":
    main()
