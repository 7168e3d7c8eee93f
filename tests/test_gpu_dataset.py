"""SURVEY.md §8(f) N4 end to end: archive tiles on disk -> native batch decode -> one copy -> GPU preprocess (standalone, or inside
SNUNet's first convolution) against the per-sample Dataset path (tests/test_dataset_cpu.py pins that one on the reference's torch
expressions).  Bit-exact."""
import gzip
import json
import os
import pickle
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))

from test_dataset_cpu import TRAIN, VAL, TEST, _configs            # noqa: E402


@pytest.fixture(scope="module")
def archive(tmp_path_factory):
    from make_synthetic_archive import make
    root = str(tmp_path_factory.mktemp("ks"))
    os.makedirs(os.path.join(root, "pickle"))
    tr, _ = make(root, TRAIN, tiles_per_act=4, seed=1)
    te, _ = make(root, VAL + TEST, tiles_per_act=4, seed=2)
    pickle.dump(tr, gzip.open(os.path.join(root, "pickle", "train.gz"), "wb"))
    pickle.dump(te, gzip.open(os.path.join(root, "pickle", "test.gz"), "wb"))
    return root


@pytest.mark.parametrize("channels", [["vv", "vh"], ["vv", "vh", "vh/vv"]])
def test_batch_loader_equals_the_collated_per_sample_dataset(archive, channels):
    from kurosiwo_amd.dataset import Dataset, TileBatchLoader
    mean = [0.0953, 0.0264, 0.3][:len(channels)]
    std = [0.0427, 0.0215, 0.2][:len(channels)]
    cfg = _configs(archive, dem=True, channels=channels, data_mean=mean, data_std=std, device="cuda")
    ds = Dataset("train", cfg)
    ref = list(torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False))
    got = list(TileBatchLoader(ds, 4, device="cuda", threads=4))
    assert len(ref) == len(got) == 2
    unbuffered = list(TileBatchLoader(ds, 4, device="cuda", threads=4, prefetch=0))
    assert all(torch.equal(a[p], b[p]) for a, b in zip(got, unbuffered) for p in (2, 3, 6, 9, 10))
    it = iter(TileBatchLoader(ds, 4, device="cuda", threads=4))         # a consumer that walks away: the producer thread ends
    next(it)
    it.close()
    for r, g in zip(ref, got):
        assert len(r) == len(g) == 13
        for pos in (2, 6, 9, 10):
            assert g[pos].is_cuda and torch.equal(g[pos].cpu(), r[pos]), pos
        assert torch.equal(g[3].cpu(), r[3]) and torch.equal(g[11], r[11]) and torch.equal(g[12], r[12])
        for pos in (0, 1, 4, 5, 7, 8):
            assert all(torch.equal(a, b) for a, b in zip(g[pos], r[pos]))


def test_raw_tiles_into_snunet_equal_the_normalised_path(archive):
    """disk -> raw batch -> first convolution (clamp, NaN, Normalize and the DEM concat inside its load) == the Dataset path"""
    from kurosiwo_amd.dataset import Dataset, TileBatchLoader
    from kurosiwo_amd.snunet import SNUNet_ECAM
    from kurosiwo_amd.synthetic import cd_inputs
    cfg = _configs(archive, dem=True, device="cuda")
    ds = Dataset("train", cfg)
    norm = next(iter(torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)))
    raw = next(iter(TileBatchLoader(ds, 4, device="cuda", raw=True)))
    torch.manual_seed(0)
    model = SNUNet_ECAM(3, 3, base_channel=16, precision="bf16").cuda().train()

    def step(fn):
        for p in model.parameters():
            p.grad = None
        out = fn()
        torch.nn.functional.cross_entropy(out, norm[3].cuda(), ignore_index=3).backward()
        return out.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]
    (xa, xb), _ = cd_inputs(norm, ("pre_event_1", "post_event"), True)
    out0, g0 = step(lambda: model(xa.cuda(), xb.cuda()))
    # the DEM arrives normalised from the loader (its gap filling is host work): mean 0 / std 1 / no clamp for that channel
    model.set_input_pipeline(cfg["data_mean"] + [0.0], cfg["data_std"] + [1.0], [cfg["clamp_input"]] * 2 + [-1.0])
    out1, g1 = step(lambda: model(raw[6], raw[2], raw[10]))
    assert torch.isnan(raw[2]).any() and torch.isfinite(out1).all()
    assert torch.equal(out0, out1)
    assert all(torch.equal(a, b) for a, b in zip(g0, g1))


def test_main_entry_trains_from_the_archive(archive, tmp_path, monkeypatch):
    """main.py with the reference's flags, data config pointed at an archive on disk: loaders -> trainer -> checkpoint -> test"""
    import shutil
    import main as entry
    from kurosiwo_amd.config import load_json5
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copytree(os.path.join(root, "configs"), tmp_path / "configs")
    dc = load_json5(tmp_path / "configs" / "train" / "data_config.json")
    dc.update(train_acts=TRAIN, val_acts=VAL, test_acts=TEST, train_pickle=os.path.join(archive, "pickle", "train.gz"),
              test_pickle=os.path.join(archive, "pickle", "test.gz"))
    json.dump(dc, open(tmp_path / "configs" / "train" / "data_config.json", "w"))
    cc = load_json5(tmp_path / "configs" / "config.json")
    cc["root_path"] = archive
    json.dump(cc, open(tmp_path / "configs" / "config.json", "w"))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("KSMI_DATA", "archive")
    seen = {}
    import kurosiwo_amd.dataset as DS
    orig = DS.TileBatchLoader._load

    def spy(self, idx):
        seen["batches"] = seen.get("batches", 0) + 1
        return orig(self, idx)
    monkeypatch.setattr(DS.TileBatchLoader, "_load", spy)
    from kurosiwo_amd.snunet import SNUNet_ECAM
    orig_set = SNUNet_ECAM.set_input_pipeline

    def spy_set(self, *a, **k):
        seen["fused"] = seen.get("fused", 0) + 1
        return orig_set(self, *a, **k)
    monkeypatch.setattr(SNUNet_ECAM, "set_input_pipeline", spy_set)
    argv = ["--method", "snunet", "--inputs", "pre_event_1", "post_event", "--batch_size", "4", "--dem"]
    miou = entry.main(argv)
    assert 0.0 <= miou <= 100.0 and seen["batches"] >= 2 + 1 + 1 and seen["fused"] >= 3     # train, validation, test
    assert list((tmp_path / "checkpoints" / "snunet").glob("*/best_segmentation.pt"))
    # the same run with the loaders normalising (ksmi_sar_preprocess) and the trainer concatenating the DEM: the same arithmetic
    monkeypatch.setenv("KSMI_FUSE_INPUT", "0")
    seen["fused"] = 0
    import time
    time.sleep(1.1)                                                          # (checkpoint folders are stamped to the second)
    assert entry.main(argv) == miou and seen["fused"] == 0


def test_segmentation_entry_trains_from_the_archive(archive, tmp_path, monkeypatch):
    """the segmentation route (main.py --method unet: segmentation_trainer.py:54-171) on the same loaders: [post, dem, pre1, pre2] concat
    of device tensors coming from the batch loader"""
    import shutil
    import main as entry
    from kurosiwo_amd.config import load_json5
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copytree(os.path.join(root, "configs"), tmp_path / "configs")
    dc = load_json5(tmp_path / "configs" / "train" / "data_config.json")
    dc.update(train_acts=TRAIN, val_acts=VAL, test_acts=TEST, train_pickle=os.path.join(archive, "pickle", "train.gz"),
              test_pickle=os.path.join(archive, "pickle", "test.gz"))
    json.dump(dc, open(tmp_path / "configs" / "train" / "data_config.json", "w"))
    cc = load_json5(tmp_path / "configs" / "config.json")
    cc["root_path"] = archive
    json.dump(cc, open(tmp_path / "configs" / "config.json", "w"))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("KSMI_DATA", "archive")
    import kurosiwo_amd.dataset as DS
    seen = {"n": 0}
    orig = DS.TileBatchLoader._load

    def spy(self, idx):
        seen["n"] += 1
        return orig(self, idx)
    monkeypatch.setattr(DS.TileBatchLoader, "_load", spy)
    miou = entry.main(["--method", "unet", "--batch_size", "4", "--dem"])
    assert 0.0 <= miou <= 100.0 and seen["n"] >= 4


def test_slc_batch_loader_and_changeformer_from_an_slc_archive(tmp_path, monkeypatch):
    """4-band SLC tiles (BASELINE.json configs[3]): the batch loader (normalised on the GPU, DEM gaps filled natively) equals the collated
    per-sample SLCDataset bit for bit; main.py --method changeformer trains from the SLC archive"""
    import shutil
    import main as entry
    from make_synthetic_archive import make
    from kurosiwo_amd.config import load_json5
    from kurosiwo_amd.dataset import SLCDataset, TileBatchLoader
    root = str(tmp_path / "slc")
    os.makedirs(os.path.join(root, "pickle"))
    tr, _ = make(root, TRAIN, tiles_per_act=4, seed=5, slc=True)
    te, _ = make(root, VAL + TEST, tiles_per_act=4, seed=6, slc=True)
    json.dump(tr, open(os.path.join(root, "pickle", "train.json"), "w"))
    json.dump(te, open(os.path.join(root, "pickle", "test.json"), "w"))
    slc = dict(slc=True, slc_root_path=root, train_json=os.path.join(root, "pickle", "train.json"), test_json=os.path.join(root, "pickle", "test.json"),
               slc_mean=[0.022367, 39.242, 81.13, 0.043526], slc_std=[1.2843, 25.6152, 58.0151, 1.2844], slc_dem_mean=82.96, slc_dem_std=153.71)
    cfg = _configs(root, dem=True, device="cuda", **slc)
    ds = SLCDataset("train", cfg)
    ref = list(torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False))
    got = list(TileBatchLoader(ds, 4, device="cuda", threads=4))
    assert len(ref) == len(got) == 2
    for r, g in zip(ref, got):
        assert len(r) == len(g) == 13
        for pos in (2, 6, 9, 10):
            assert g[pos].is_cuda and torch.equal(g[pos].cpu(), r[pos]), pos
        assert torch.equal(g[3].cpu(), r[3]) and torch.equal(g[11], r[11]) and torch.equal(g[12], r[12])
    # end to end
    work = tmp_path / "run"
    shutil.copytree(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs"), work / "configs")
    dc = load_json5(work / "configs" / "train" / "data_config.json")
    dc.update(train_acts=TRAIN, val_acts=VAL, test_acts=TEST, **{k: v for k, v in slc.items() if k not in ("slc_dem_mean", "slc_dem_std")})
    json.dump(dc, open(work / "configs" / "train" / "data_config.json", "w"))
    monkeypatch.chdir(work)
    monkeypatch.setenv("KSMI_DATA", "archive")
    import kurosiwo_amd.dataset as DS
    seen = {"n": 0}
    orig = DS.TileBatchLoader._load_slc

    def spy(self, idx):
        seen["n"] += 1
        return orig(self, idx)
    monkeypatch.setattr(DS.TileBatchLoader, "_load_slc", spy)
    miou = entry.main(["--method", "changeformer", "--inputs", "pre_event_1", "post_event", "--batch_size", "4"])
    assert 0.0 <= miou <= 100.0 and seen["n"] >= 4
