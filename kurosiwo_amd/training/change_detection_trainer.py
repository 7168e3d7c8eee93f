"""Change-detection trainer with the reference's entry points and return contract
(/root/reference/training/change_detection_trainer.py: train_change_detection :18-322,
eval_change_detection :325-791), on the fused HIP train step.

Kept: batch-tuple unpacking, input assembly from configs['inputs'] (+dem), Adam/SGD choice, per-epoch
LR schedule step, checkpoint dict format + best_segmentation.{pt,txt}, metric definitions and the
`(100*acc[4], 100*meanF1[:3], 100*mIoU)` return value, per-AOI / water-only metrics.
Changed on purpose: no per-step host syncs -- losses and the 4x4 confusion
matrix accumulate on the device and are read once per logging interval; wandb / kornia image
panels are dropped.
"""
from pathlib import Path

import torch

from .. import distributed as D
from ..config import init_lr_scheduler
from ..loss import create_loss
from ..metrics import ConfusionMetrics, GroupedConfusion, metrics_from_cm
from ..optim import FusedAdam, FusedAdamW, FusedSGD
from ..synthetic import cd_inputs
from ..trainer import CDTrainStep

CLASS_LABELS = {0: "No water", 1: "Permanent Waters", 2: "Floods", 3: "Invalid pixels"}


def multi_scale_prediction(outputs):
    """change_detection_trainer.py:139-146 (`multi_scale_infer`: train-time metric predictions only; the evaluation loop :394-395 always
    takes output[-1]): the mean of the five ChangeFormer outputs, the coarse ones resized to the last one's size by nearest neighbour."""
    size = outputs[-1].shape[2]
    final = torch.zeros_like(outputs[-1])
    for pred in outputs:
        final = final + (torch.nn.functional.interpolate(pred, size=size, mode="nearest") if pred.shape[2] != size else pred)
    return final / len(outputs)


def _make_optimizer(model, configs, model_configs):
    if configs["method"] in ("bit-cd", "hfa-net") or model_configs.get("optimizer") == "sgd":
        return FusedSGD(model.parameters(), lr=model_configs["learning_rate"], momentum=model_configs.get("momentum", 0.0),
                        weight_decay=model_configs.get("weight_decay", 0.0))
    if model_configs["optimizer"] == "adam":
        # the reference ignores betas / weight_decay of the method json for Adam (cd_trainer:52-54)
        return FusedAdam(model.parameters(), lr=model_configs["learning_rate"])
    if model_configs["optimizer"] == "adamw":                # cd_trainer:55-60
        return FusedAdamW(model.parameters(), lr=model_configs["learning_rate"], betas=tuple(model_configs["betas"]),
                          weight_decay=model_configs["weight_decay"])
    raise NotImplementedError(model_configs["optimizer"])


def _print_metrics(prefix, m, loss):
    if not D.is_main():
        return
    print(f"{prefix} loss {loss:.5f} | mIoU {100 * float(m['miou']):.2f} | "
          + " | ".join(f"{CLASS_LABELS[c]}: acc {100 * float(m['accuracy'][c]):.2f} F1 {100 * float(m['f1'][c]):.2f} "
                       f"IoU {100 * float(m['iou'][c]):.2f}" for c in range(3)))


def fuse_input_pipeline(model, loaders, configs):
    """with archive tiles coming from dataset.TileBatchLoader and a model that normalises inside its first
    convolution (SNUNet_ECAM.set_input_pipeline), the loaders hand out RAW tiles and clamp / nan_to_num / Normalize / the DEM concat
    (dataset/Dataset.py:164-168,193-198; the torch.cat of :117-133 here) never exist as separate passes.  Returns True when on;
    configs["fuse_input_pipeline"] = false or KSMI_FUSE_INPUT=0 keeps the loaders normalising (ksmi_sar_preprocess)."""
    import os
    from ..dataset import TileBatchLoader
    if not (hasattr(model, "set_input_pipeline") and all(isinstance(ld, TileBatchLoader) and not ld.slc for ld in loaders)
            and configs.get("fuse_input_pipeline", True) and os.environ.get("KSMI_FUSE_INPUT", "1") != "0"):
        return False
    ndem = 1 if configs["dem"] else 0        # the DEM arrives standardised (its gap filling is host work): identity for that channel
    model.set_input_pipeline(list(configs["data_mean"]) + [0.0] * ndem, list(configs["data_std"]) + [1.0] * ndem,
                             [configs["clamp_input"]] * len(configs["data_mean"]) + [-1.0] * ndem)
    for ld in loaders:
        ld.raw = True
    return True


def _eval_fusion(model, loader, configs):
    """evaluation may run on a freshly loaded model (main.py: best checkpoint -> test): settle model and loader on the same side"""
    fused = fuse_input_pipeline(model, (loader,), configs)
    if not fused and hasattr(loader, "raw"):
        loader.raw = False
    return fused


def _fused_inputs(batch, configs):
    """cd_inputs without the concat: (xA, xB, dem or None), mask"""
    (xA, xB), mask = cd_inputs(batch, configs["inputs"], False)
    return (xA, xB, batch[10] if configs["dem"] else None), mask


def train_change_detection(model, train_loader, val_loader, test_loader, configs, model_configs):
    assert len(configs["inputs"]) == 2, f'Model {model_configs["method"]} requires exactly 2 input images.'
    dev = torch.device(configs["device"])
    model.to(dev)
    fused = fuse_input_pipeline(model, (train_loader, val_loader, test_loader), configs)
    D.broadcast_model_(model)                       # identical weights / BatchNorm statistics on every rank (rank 0's)
    main = D.is_main()
    optimizer = _make_optimizer(model, configs, model_configs)
    lr_scheduler = init_lr_scheduler(optimizer, configs, model_configs, steps=len(train_loader))
    metrics = ConfusionMetrics(dev)
    best_val, loss_val = 0.0, float("nan")
    step = None
    if main:
        print(f'===== checkpoint_path: {configs["checkpoint_path"]} ====')
    for epoch in range(0, configs["epochs"]):
        model.train()
        D.set_loader_epoch(train_loader, epoch)
        metrics.reset()
        loss_acc = torch.zeros(3, dtype=torch.float32, device=dev)
        nb = 0
        for index, batch in enumerate(train_loader):
            batch = D.shard_batch(batch)            # this rank's contiguous slice of the global batch (§8(e))
            if fused:
                (xA, xB, xdem), mask = _fused_inputs(batch, configs)
            else:
                ((xA, xB), mask), xdem = cd_inputs(batch, configs["inputs"], bool(configs["dem"])), None
            if step is None or step.B != xA.shape[0]:
                step = CDTrainStep(model, xA.shape[0], xA.shape[2], xA.shape[3], configs["loss_function"],
                                   configs.get("class_weights", [1.0, 1.0, 1.0]), optimizer=optimizer, graph=configs.get("hip_graph", False),
                                   overlap_wgrad=configs.get("overlap_wgrad", True), grad_dtype=configs.get("dp_grad_dtype"), dp_mode=configs.get("dp_mode"), overlap_lanes=configs.get("overlap_lanes", True),
                                   **({"tail": xdem.shape[1]} if xdem is not None else {}))
            ins = (xA, xB) + ((xdem,) if xdem is not None else ())
            step.step(*[t.to(dev, non_blocking=True) for t in ins], mask.to(dev, non_blocking=True))
            if configs["method"] == "changeformer" and model_configs.get("multi_scale_infer"):
                metrics.update(multi_scale_prediction(step.plan.outputs), step.labels)
            else:
                metrics.update(step.plan.logits, step.labels)
            loss_acc += step.loss_out
            nb += 1
            if configs.get("on_screen_prints") and (index + 1) % configs["print_frequency"] == 0:
                _print_metrics(f"[{epoch}:{index + 1}]", metrics.compute(), float(loss_acc[0]) / nb)
        D.all_reduce_sum_(metrics.cm, loss_acc)
        loss_val = float(loss_acc[0]) / max(nb, 1) / D.world_size()
        _print_metrics(f"Epoch {epoch} train", metrics.compute(), loss_val)
        if main and (epoch + 1) % configs.get("train_save_checkpoint_freq", 1) == 0:
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                        "lr_scheduler_state_dict": lr_scheduler.state_dict(), "loss": loss_val, "rng_state": _rng_words(model)},
                       Path(configs["checkpoint_path"]) / f"checkpoint_epoch={epoch}.pt")
        lr_scheduler.step()
        val_acc, val_score, miou = eval_change_detection(model, val_loader, settype="Validation", configs=configs,
                                                         model_configs=model_configs)
        improved = miou > best_val                   # strict, as change_detection_trainer.py:305 (a tie keeps the earlier checkpoint)
        if improved:
            best_val = miou
        if improved and main:                        # (every rank holds the same all-reduced metrics; rank 0 writes)
            print(f"New best validation mIoU: {miou}")
            print(f'Saving model to: {configs["checkpoint_path"]}/best_segmentation.pt')
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                        "lr_scheduler_state_dict": lr_scheduler.state_dict(), "loss": loss_val, "rng_state": _rng_words(model)},
                       Path(configs["checkpoint_path"]) / "best_segmentation.pt")
            with open(Path(configs["checkpoint_path"]) / "best_segmentation.txt", "w") as f:
                f.write(f"{epoch}\n")
                f.write(f"{miou}")
        D.barrier()                                 # checkpoints of this epoch are on disk before any rank moves on


def _rng_words(model):
    """{seed, step} of the counter-based Dropout / DropPath stream (a new key next to the reference's checkpoint keys: a resumed run
    continues the stream instead of replaying the masks of the first steps)"""
    return [int(v) & 0xFFFFFFFF for v in model.rng_state().cpu().tolist()] if hasattr(model, "rng_state") else None


def eval_change_detection(model, loader, settype, configs=None, model_configs=None):
    dev = torch.device(configs["device"])
    grouped = GroupedConfusion(dev, [list(getattr(loader.dataset, "activations", [])) if configs.get("log_AOI_metrics") else [],
                                     [1, 2, 3] if configs.get("log_zone_metrics") else []])
    metrics, (per_aoi, per_zone) = grouped.total, grouped.groups
    criterion = create_loss(configs, mode="val")
    model.to(dev)
    model.eval()
    fused = _eval_fusion(model, loader, configs)
    total_loss = torch.zeros((), dtype=torch.float32, device=dev)
    nsamples = 0
    with torch.no_grad():
        for batch in loader:
            batch = D.shard_batch(batch, even=False)
            if len(batch) == 0:        # this rank's slice of a ragged last batch is empty (n % batch < world): nothing to evaluate,
                continue               # the all-reduce below still runs once on every rank
            if fused:
                (xA, xB, xdem), mask = _fused_inputs(batch, configs)
            else:
                ((xA, xB), mask), xdem = cd_inputs(batch, configs["inputs"], bool(configs["dem"])), None
            if xA.shape[0] == 0:
                continue
            xA, xB, mask = xA.to(dev), xB.to(dev), mask.to(dev)
            clz, activ = batch[-2], batch[-1]
            output = model(xA, xB) if xdem is None else model(xA, xB, xdem.to(dev))
            if configs["method"] == "changeformer":
                output = output[-1]
            total_loss += criterion(output, mask) * xA.size(0)
            nsamples += xA.size(0)
            # overall + per-AOI + per-zone confusion matrices in ONE launch (metrics.GroupedConfusion): the group keys of the batch are
            # host data, turned into device slot indices without a synchronisation
            grouped.update(output, mask, (activ if per_aoi else None, clz if per_zone else None))
    ns = torch.tensor([float(nsamples)], dtype=torch.float64, device=dev)
    D.all_reduce_sum_(metrics.cm, total_loss, ns, *[g.cm for g in list(per_aoi.values()) + list(per_zone.values())])
    nsamples = int(ns.item())
    m = metrics.compute()
    loss = float(total_loss) / max(nsamples, 1)
    _print_metrics(f"{settype}", m, loss)
    if configs.get("evaluate_water") and D.is_main():
        cm = metrics.cm.cpu().double()
        # water-only F1: classes {1,2} merged (cd_trainer:414-419)
        w = torch.zeros((4, 4), dtype=torch.float64)
        w[0, 0] = cm[0, 0]
        w[0, 1] = cm[0, 1] + cm[0, 2]
        w[1, 0] = cm[1, 0] + cm[2, 0]
        w[1, 1] = cm[1:3, 1:3].sum()
        wm = metrics_from_cm(w)
        print(f'{settype} water-only F1: no-water {100 * float(wm["f1"][0]):.2f} water {100 * float(wm["f1"][1]):.2f}')
    for name, group in (("AOI", per_aoi), ("climate zone", per_zone)):
        for k, cmx in group.items():
            if int(cmx.cm.sum()) > 0 and D.is_main():
                gm = cmx.compute()
                print(f"{settype} {name} {k}: mIoU {100 * float(gm['miou']):.2f}")
    return 100 * m["accuracy"], 100 * m["f1"][:3].mean(), 100 * m["miou"]
