"""the evaluation loop + metric suite of the trainers (eval_change_detection: per-class accuracy / F1 / IoU from the
4x4 confusion matrix, water-only F1, ragged last batch) pinned to an oracle computation: the CPU oracle's eval-mode forward
(oracle/snunet_ref.py) -> argmax -> oracle/metrics_ref.py on the same synthetic validation tiles.
Reference: training/change_detection_trainer.py:325-419 (loop), :414-419 (water-only), :532,791 (returned triple)."""
import os
import re
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_eval_change_detection_returns_the_oracle_metrics(tmp_path, monkeypatch, capsys):
    from kurosiwo_amd.config import load_json5, update_config
    from kurosiwo_amd.data import prepare_loaders
    from kurosiwo_amd.snunet import SNUNet_ECAM
    from kurosiwo_amd.synthetic import cd_inputs
    from kurosiwo_amd.training.change_detection_trainer import eval_change_detection
    from oracle import metrics_ref, snunet_ref as R
    from oracle.seeded import seeded_fill_
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copytree(os.path.join(root, "configs"), tmp_path / "configs")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("KSMI_SYNTHETIC_TILES", "4,6,4")                 # 6 validation tiles, batch 4: the last batch is ragged
    configs = load_json5("configs/config.json")
    model_configs = load_json5("configs/method/snunet/snunet.json")
    configs.update(model_configs)
    configs.update(task="cd", method="snunet")           # (the shipped defaults are the reference's: segmentation / unet)

    class A:
        inputs, dem, slope = ["pre_event_1", "post_event"], False, False
    configs = update_config(configs, A)
    configs.update(batch_size=4, precision="fp32", evaluate_water=True, log_AOI_metrics=True, log_zone_metrics=True, device="cuda:0")
    _, val_loader, _ = prepare_loaders(configs)
    sd = seeded_fill_(R.new_state_dict(2, 3, 32))
    model = SNUNet_ECAM(2, 3, base_channel=32, precision="fp32")
    model.load_state_dict(sd)
    capsys.readouterr()
    acc, f1, miou = eval_change_detection(model, val_loader, settype="Validation", configs=configs, model_configs=model_configs)
    printed = capsys.readouterr().out
    # ---- oracle: same tiles, same order
    cm = np.zeros((4, 4), np.int64)
    per_aoi, per_zone = {}, {}
    for batch in val_loader:
        (xA, xB), mask = cd_inputs(batch, configs["inputs"], False)
        with torch.no_grad():
            logits = R.snunet_forward(sd, xA, xB, training=False).numpy()
        pred = metrics_ref.argmax_lowest_index(logits)
        cm += metrics_ref.confusion_matrix(pred, mask.numpy())
        for i, a in enumerate(batch[-1].tolist()):
            per_aoi[a] = per_aoi.get(a, 0) + metrics_ref.confusion_matrix(pred[i], mask[i].numpy())
        for i, z in enumerate(batch[-2].tolist()):
            per_zone[z] = per_zone.get(z, 0) + metrics_ref.confusion_matrix(pred[i], mask[i].numpy())
    ref = metrics_ref.metrics_from_cm(cm)
    # fp32 logits agree to ~1e-6: a handful of near-tie pixels may flip -> a few 1e-4 percentage points
    assert np.abs(acc.numpy() - 100 * ref["accuracy"]).max() < 2e-3
    assert abs(float(f1) - 100 * ref["f1"][:3].mean()) < 2e-3
    assert abs(float(miou) - 100 * ref["miou"]) < 2e-3
    # water-only F1 (cd_trainer:414-419): classes {1, 2} merged
    w = np.zeros((4, 4), np.int64)
    w[0, 0], w[0, 1], w[1, 0], w[1, 1] = cm[0, 0], cm[0, 1] + cm[0, 2], cm[1, 0] + cm[2, 0], cm[1:3, 1:3].sum()
    wm = metrics_ref.metrics_from_cm(w)
    mo = re.search(r"Validation water-only F1: no-water ([0-9.]+) water ([0-9.]+)", printed)
    assert mo, printed
    assert abs(float(mo.group(1)) - 100 * wm["f1"][0]) < 6e-3 and abs(float(mo.group(2)) - 100 * wm["f1"][1]) < 6e-3
    # per-AOI breakdown (cd_trainer:331-337,437-472)
    for a, c in per_aoi.items():
        mo = re.search(rf"Validation AOI {a}: mIoU ([0-9.]+)", printed)
        assert mo, (a, printed)
        assert abs(float(mo.group(1)) - 100 * metrics_ref.metrics_from_cm(c)["miou"]) < 6e-3
    for z, c in per_zone.items():
        if z in (1, 2, 3) and int(c.sum()) > 0:
            mo = re.search(rf"Validation climate zone {z}: mIoU ([0-9.]+)", printed)
            assert mo, (z, printed)
            assert abs(float(mo.group(1)) - 100 * metrics_ref.metrics_from_cm(c)["miou"]) < 6e-3


def test_grouped_confusion_one_launch_equals_per_sample_oracle():
    """ksmi_argmax_confusion_grouped (metrics.GroupedConfusion): the overall matrix and two families of per-group matrices from ONE launch
    per batch equal the integer oracle applied sample by sample (the reference keeps one torchmetrics object per activation id / zone:
    cd_trainer:331-337, 437-472); keys that name no group, ties (lowest index wins) and ignored pixels included.  Bit-exact (integers)."""
    from kurosiwo_amd.metrics import GroupedConfusion
    from oracle import metrics_ref
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    B, C, H, W = 7, 3, 33, 41
    aoi_keys, zone_keys = [130, 321, 470], [1, 2, 3]
    gc = GroupedConfusion(dev, [aoi_keys, zone_keys])
    want_total = np.zeros((4, 4), np.int64)
    want_a = {k: np.zeros((4, 4), np.int64) for k in aoi_keys}
    want_z = {k: np.zeros((4, 4), np.int64) for k in zone_keys}
    for it in range(3):
        logits = torch.randn((B, C, H, W), generator=g)
        logits[:, :, ::5, ::3] = 0.25                                           # ties across all classes: argmax = class 0
        mask = torch.randint(0, 4, (B, H, W), generator=g)
        activ = torch.tensor([130, 321, 999, 470, 130, 130, 321])               # 999: no such group
        clz = torch.tensor([1, 2, 3, 0, 2, 2, 1])                               # 0: no such zone
        gc.update(logits.to(dev), mask.to(dev), (activ, clz))
        pred = metrics_ref.argmax_lowest_index(logits.numpy())
        for i in range(B):
            c = metrics_ref.confusion_matrix(pred[i], mask[i].numpy())
            want_total += c
            if int(activ[i]) in want_a:
                want_a[int(activ[i])] += c
            if int(clz[i]) in want_z:
                want_z[int(clz[i])] += c
    assert np.array_equal(gc.total.cm.cpu().numpy(), want_total)
    for k in aoi_keys:
        assert np.array_equal(gc.groups[0][k].cm.cpu().numpy(), want_a[k]), k
    for k in zone_keys:
        assert np.array_equal(gc.groups[1][k].cm.cpu().numpy(), want_z[k]), k
    # one family only / none: the same entry point
    gc1 = GroupedConfusion(dev, [[], zone_keys])
    gc1.update(logits.to(dev), mask.to(dev), (None, clz))
    assert int(gc1.total.cm.sum()) == int((mask != 3).sum()) and int(gc1.tables[1].sum()) == int((mask[clz != 0] != 3).sum())
