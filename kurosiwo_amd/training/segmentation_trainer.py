"""Segmentation trainer with the reference's entry points and return contract
(/root/reference/training/segmentation_trainer.py: train_semantic_segmentation :16-255,
eval_semantic_segmentation :258-1011) for the FloodViT path, on the fused HIP train step.

Kept: batch-tuple unpacking, the channel concat [post, (dem), pre1, pre2] from configs['inputs'] (:107-147), Adam
(:36), per-epoch LR schedule step (:231), weighted-CE train / unweighted val criterion (create_loss), metric
definitions, best-model selection on validation mIoU, `(100*acc, 100*meanF1[:3], 100*mIoU)` return value.
Changed on purpose: no per-step host syncs; wandb / kornia panels dropped; the best model is
saved as a state-dict bundle next to the whole-module pickle the reference writes (:255) so both loaders work.
"""
from pathlib import Path

import torch

from .. import distributed as D
from ..config import init_lr_scheduler
from ..loss import create_loss
from ..metrics import ConfusionMetrics, GroupedConfusion, metrics_from_cm
from ..optim import FusedAdam
from ..synthetic import seg_inputs
from ..trainer import SegTrainStep
from .change_detection_trainer import _print_metrics


def train_semantic_segmentation(model, train_loader, val_loader, test_loader, configs, model_configs):
    dev = torch.device(configs["device"])
    model.to(dev)
    D.broadcast_model_(model)
    main = D.is_main()
    optimizer = FusedAdam(model.parameters(), lr=model_configs["learning_rate"])
    lr_scheduler = init_lr_scheduler(optimizer, configs, model_configs, steps=len(train_loader))
    metrics = ConfusionMetrics(dev)
    best_val, best_stats = 0.0, {}
    step = None
    weights = configs.get("class_weights", [1.0, 1.0, 1.0])
    for epoch in range(configs["epochs"]):
        model.train()
        D.set_loader_epoch(train_loader, epoch)
        metrics.reset()
        loss_acc = torch.zeros(3, dtype=torch.float32, device=dev)
        nb = 0
        for index, batch in enumerate(train_loader):
            batch = D.shard_batch(batch)
            image, mask = seg_inputs(batch, configs["inputs"], bool(configs["dem"]))
            if step is None or step.B != image.shape[0]:
                step = SegTrainStep(model, image.shape[0], configs["loss_function"], weights, optimizer=optimizer, graph=configs.get("hip_graph", False),
                                    overlap_wgrad=configs.get("overlap_wgrad", True), grad_dtype=configs.get("dp_grad_dtype"), dp_mode=configs.get("dp_mode"), **({} if hasattr(model, "hp") else {"image_size": tuple(image.shape[2:])}))
            step.step(image.to(dev, non_blocking=True), mask.to(dev, non_blocking=True))
            metrics.update(step.plan.logits, step.labels)
            loss_acc += step.loss_out
            nb += 1
            if configs.get("on_screen_prints") and (index + 1) % configs["print_frequency"] == 0:
                _print_metrics(f"[{epoch}:{index + 1}]", metrics.compute(), float(loss_acc[0]) / nb)
        D.all_reduce_sum_(metrics.cm, loss_acc)
        _print_metrics(f"Epoch {epoch} train", metrics.compute(), float(loss_acc[0]) / max(nb, 1) / D.world_size())
        lr_scheduler.step()
        model.eval()
        val_acc, val_score, miou = eval_semantic_segmentation(model, val_loader, settype="Val", configs=configs,
                                                              model_configs=model_configs)
        improved = miou > best_val                   # strict, as segmentation_trainer.py:243 (a tie keeps the earlier checkpoint)
        if improved:
            best_val = miou
            best_stats["miou"], best_stats["epoch"] = best_val, epoch
        if improved and main:
            print("Epoch: ", epoch)
            print("New best validation mIOU: ", float(miou))
            print("Saving model to: ", configs["checkpoint_path"] + "/" + "best_segmentation.pt")
            model._plans = {}                 # plans hold device buffers and ctypes descriptors: not part of the pickle
            torch.save(model, Path(configs["checkpoint_path"]) / "best_segmentation.pt")
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict()},
                       Path(configs["checkpoint_path"]) / "best_segmentation_state.pt")
        D.barrier()
    return best_stats


def eval_semantic_segmentation(model, loader, configs=None, settype="Test", model_configs=None):
    dev = torch.device(configs["device"])
    grouped = GroupedConfusion(dev, [list(getattr(loader.dataset, "activations", [])) if configs.get("log_AOI_metrics") else [],
                                     [1, 2, 3] if configs.get("log_zone_metrics") else []])
    metrics, (per_aoi, per_zone) = grouped.total, grouped.groups
    criterion = create_loss(configs, mode="val")
    model.to(dev)
    model.eval()
    total_loss = torch.zeros((), dtype=torch.float32, device=dev)
    nsamples = 0
    with torch.no_grad():
        for batch in loader:
            batch = D.shard_batch(batch, even=False)
            if len(batch) == 0:        # empty slice of a ragged last batch on this rank (see eval_change_detection)
                continue
            image, mask = seg_inputs(batch, configs["inputs"], bool(configs["dem"]))
            if image.shape[0] == 0:
                continue
            image, mask = image.to(dev), mask.to(dev)
            clz, activ = batch[-2], batch[-1]
            output = model(image)
            total_loss += criterion(output, mask) * image.size(0)
            nsamples += image.size(0)
            # overall + per-AOI + per-zone confusion matrices in ONE launch (metrics.GroupedConfusion): the group keys of the batch are
            # host data, turned into device slot indices without a synchronisation
            grouped.update(output, mask, (activ if per_aoi else None, clz if per_zone else None))
    ns = torch.tensor([float(nsamples)], dtype=torch.float64, device=dev)
    D.all_reduce_sum_(metrics.cm, total_loss, ns, *[g.cm for g in list(per_aoi.values()) + list(per_zone.values())])
    nsamples = int(ns.item())
    m = metrics.compute()
    _print_metrics(f"{settype}", m, float(total_loss) / max(nsamples, 1))
    if configs.get("evaluate_water") and D.is_main():
        cm = metrics.cm.cpu().double()
        w = torch.zeros((4, 4), dtype=torch.float64)
        w[0, 0], w[0, 1], w[1, 0], w[1, 1] = cm[0, 0], cm[0, 1] + cm[0, 2], cm[1, 0] + cm[2, 0], cm[1:3, 1:3].sum()
        wm = metrics_from_cm(w)
        print(f'{settype} water-only F1: no-water {100 * float(wm["f1"][0]):.2f} water {100 * float(wm["f1"][1]):.2f}')
    for name, group in (("AOI", per_aoi), ("climate zone", per_zone)):
        for k, cmx in group.items():
            if int(cmx.cm.sum()) > 0 and D.is_main():
                print(f"{settype} {name} {k}: mIoU {100 * float(cmx.compute()['miou']):.2f}")
    return 100 * m["accuracy"], 100 * m["f1"][:3].mean(), 100 * m["miou"]
