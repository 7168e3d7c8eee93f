"""MAE pre-training loop with the reference's entry points (/root/reference/training/train_mae.py: adjust_learning_rate :14-33,
train_epoch :41-123, train :130-229) on the HIP MAE step (kurosiwo_amd/mae.py).

Kept: per-iteration half-cycle cosine schedule with linear warm-up in (fractional) epochs (:14-33, called on the update iterations
only, :76-79), learning rate scaled by `accumulate_gradients` (:160), the reference's accumulation pattern AS WRITTEN -- every
batch is forwarded, but zero_grad / backward / optimizer.step run only on iterations where (idx + 1) % accumulate == 0 or on the last
one, with the loss divided by `accumulate_gradients` (:65-70, :108-122) -- Adam, checkpoints `mae_{epoch}.pt` (state dict),
`vit_{epoch}.pt` / `trained_vit_{epochs}.pt` (pickled encoder modules the fine-tuning config points at) and
`mae_vit_{epochs}.pt` (encoder state dict) (:203-229).
Changed on purpose: bf16 compute replaces torch.cuda.amp autocast + GradScaler (bf16 needs no loss scaling); the running loss is
accumulated on the device and read once per logging interval; wandb dropped; the SSL dataset reader is out of scope
(DESIGN.md §7), `loader` yields (image, ...) tuples and a synthetic one is built when none is passed.
"""
import math
import os

import torch

from ..floodvit import ViT
from ..mae import build_mae
from ..optim import FusedAdam


def adjust_learning_rate(optimizer, epoch, configs):
    """Decay the learning rate with half-cycle cosine after warmup (train_mae.py:14-33)."""
    if epoch < configs["warmup_epochs"]:
        lr = configs["lr"] * epoch / configs["warmup_epochs"]
    else:
        lr = configs["min_lr"] + (configs["lr"] - configs["min_lr"]) * 0.5 * (
            1.0 + math.cos(math.pi * (epoch - configs["warmup_epochs"]) / (configs["epochs"] - configs["warmup_epochs"])))
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr


def get_current_learning_rate(optimizer):
    for group in optimizer.param_groups:
        return group["lr"]


def encoder_module(mae):
    """The pre-trained encoder as a stand-alone, picklable `ViT` (what the reference saves with torch.save(model.encoder))."""
    hp = mae.hp
    enc = ViT(image_size=hp["image_size"], patch_size=hp["patch_size"], num_classes=hp.get("num_classes") or 1000, dim=hp["dim"],
              depth=hp["depth"], heads=hp["heads"], mlp_dim=hp["mlp_dim"], channels=hp["channels"], dim_head=hp["dim_head"])
    sd = {k[8:]: v.detach().cpu() for k, v in mae.state_dict().items() if k.startswith("encoder.")}
    enc.load_state_dict(sd)
    return enc


def train_epoch(loader, mae, optimizer, epoch, configs, scaler=None):
    mae.train()
    configs["num_steps_per_epoch"] = configs["num_samples_per_epoch"] // configs["batch_size"]
    steps = configs["num_steps_per_epoch"]
    device = configs.get("device", configs.get("gpu"))
    accumulate = configs.get("accumulate_gradients")
    running = torch.zeros((), dtype=torch.float32, device=device)
    nbatches = 0
    it = iter(loader)
    logs = []
    for idx in range(steps):
        try:
            batch = next(it)
        except StopIteration:
            it = iter(loader)
            batch = next(it)
        update = accumulate is None or (idx + 1) % accumulate == 0 or (idx + 1) == steps
        if update:
            optimizer.zero_grad(set_to_none=True)
            adjust_learning_rate(optimizer, idx / steps + epoch, configs)     # per iteration, as the official MAE does
        image = batch[0].to(device, non_blocking=True)
        if update:
            loss = mae(image)
        else:
            with torch.no_grad():                                              # the reference forwards these batches and drops them
                loss = mae(image)
        running += loss.detach()
        nbatches += 1
        if idx % 100 == 0:
            logs.append({"Epoch": epoch, "Iteration": idx, "train loss": float(running) / nbatches,
                         "Current Learning Rate": get_current_learning_rate(optimizer)})
            print(logs[-1])
            running.zero_()
            nbatches = 0
        if update:
            if accumulate is not None:
                loss = loss / accumulate
            loss.backward()
            optimizer.step()
    return logs


class _SyntheticSSL(torch.utils.data.Dataset):
    """(image, flood, pre_event_1, pre_event_2) tuples shaped like dataset/Dataset.py SSLDataset items."""

    def __init__(self, n, channels, size, seed=999):
        self.n, self.c, self.s, self.seed = n, channels, size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        img = torch.randn(self.c, self.s, self.s, generator=g).clamp_(-2.23, 5.75)
        return img, img[:1], img, img


def train(configs, loader=None, precision="bf16"):
    configs.setdefault("lr", configs["learning_rate"])
    accumulated = 1 if configs.get("accumulate_gradients") is None else configs["accumulate_gradients"]
    configs["lr"] = configs["learning_rate"] * accumulated                     # train_mae.py:157-160
    device = configs.get("device", configs.get("gpu"))
    if loader is None:
        ds = _SyntheticSSL(max(configs["batch_size"] * 4, 8), configs["num_channels"], configs["image_size"], configs.get("seed", 999))
        loader = torch.utils.data.DataLoader(ds, batch_size=configs["batch_size"], shuffle=False, drop_last=False)
    model = build_mae(configs, precision=precision, channels=configs["num_channels"]).to(device)
    optimizer = FusedAdam(model.parameters(), lr=configs["lr"])
    start = configs.get("start_epoch") or 0
    ckpt = configs["checkpoint_path"]
    os.makedirs(ckpt, exist_ok=True)
    for epoch in range(start, configs["epochs"]):
        train_epoch(loader, model, optimizer, epoch, configs)
        torch.save(model.state_dict(), os.path.join(ckpt, f"mae_{epoch}.pt"))
        torch.save(encoder_module(model), os.path.join(ckpt, f"vit_{epoch}.pt"))
    enc = encoder_module(model)
    torch.save(enc.state_dict(), os.path.join(ckpt, f"mae_vit_{configs['epochs']}.pt"))
    torch.save(enc, os.path.join(ckpt, f"trained_vit_{configs['epochs']}.pt"))
    return model
