"""the archive Dataset classes (kurosiwo_amd/dataset.py, mirroring dataset/Dataset.py) on a synthetic archive with
the real on-disk layout (tools/make_synthetic_archive.py).  The reference classes cannot be imported here (cv2, rioxarray,
albumentations, richdem, torchio, compress_pickle are not in the image), so the expected tensors are the reference's own torch
expressions (dataset/Dataset.py:164-168 clamp + nan_to_num, :193-198 Normalize, :824-860 tuple layout) applied to the arrays that were
written to disk.  Bit-exact."""
import gzip
import json
import os
import pickle
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))

MEAN, STD = [0.0953, 0.0264], [0.0427, 0.0215]
TRAIN, VAL, TEST = [101, 102], [201], [301]


def _configs(root, **over):
    c = dict(train_acts=TRAIN, val_acts=VAL, test_acts=TEST, root_path=str(root), task="cd", data_augmentations=False,
             train_pickle=os.path.join(root, "pickle", "train.gz"), test_pickle=os.path.join(root, "pickle", "test.gz"),
             oversampling=False, negative_pickle=None, track="RandomEvents", channels=["vv", "vh"], clamp_input=0.15,
             scale_input="normalize", data_mean=MEAN, data_std=STD, dem=False, slope=False, dem_mean=[93.4313], dem_std=[1410.8382],
             uint8=False, batch_size=4, num_workers=0, device="cpu")
    c.update(over)
    return c


@pytest.fixture(scope="module")
def archive(tmp_path_factory):
    from make_synthetic_archive import make
    root = str(tmp_path_factory.mktemp("ks"))
    os.makedirs(os.path.join(root, "pickle"))
    tr, truth_tr = make(root, TRAIN + [999], tiles_per_act=3, seed=1)          # activation 999 is in no split
    te, truth_te = make(root, VAL + TEST, tiles_per_act=3, seed=2)
    pickle.dump(tr, gzip.open(os.path.join(root, "pickle", "train.gz"), "wb"))
    pickle.dump(te, gzip.open(os.path.join(root, "pickle", "test.gz"), "wb"))
    return root, (tr, truth_tr), (te, truth_te)


def _ref_concat(vv, vh, channels, clamp):
    """dataset/Dataset.py:148-169"""
    if set(channels) == {"vv", "vh", "vh/vv"}:
        img = np.vstack((vv[None], vh[None], vh[None] / (vv[None] + 1e-7)))
    elif set(channels) == {"vv", "vh"}:
        img = np.vstack((vv[None], vh[None]))
    else:
        img = vh[None]
    img = torch.from_numpy(img).float()
    if clamp is not None:
        return torch.nan_to_num(torch.clamp(img, min=0.0, max=clamp), clamp)
    return torch.nan_to_num(img, 200)


def _ref_normalize(img, mean, std):
    """torchvision.transforms.functional.normalize: tensor.sub_(mean).div_(std)"""
    m = torch.as_tensor(mean, dtype=img.dtype).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=img.dtype).view(-1, 1, 1)
    return img.clone().sub_(m).div_(s)


def test_record_selection_follows_the_activation_lists(archive, capsys):
    from kurosiwo_amd.dataset import Dataset
    root, (tr, _), (te, _) = archive
    ds = {m: Dataset(m, _configs(root)) for m in ("train", "val", "test")}
    assert len(ds["train"]) == 6 and len(ds["val"]) == 3 and len(ds["test"]) == 3
    assert ds["train"].activations == set(TRAIN) and ds["val"].activations == set(VAL) and ds["test"].activations == set(TEST)
    assert ds["train"].non_valids == [999] and "Activation:  999  not in Activations" in capsys.readouterr().out
    assert sum(ds["train"].clz_stats.values()) == 6 and ds["val"].pickle_path.endswith("test.gz")
    with pytest.raises(SystemExit):
        Dataset("train", _configs(root, train_pickle=os.path.join(root, "nope.gz")))
    with pytest.raises(NotImplementedError):
        Dataset("train", _configs(root, data_augmentations=True))


@pytest.mark.parametrize("channels", [["vv", "vh"], ["vv", "vh", "vh/vv"], ["vh"]])
def test_getitem_is_the_reference_pipeline_bit_for_bit(archive, channels):
    from kurosiwo_amd.dataset import Dataset
    root, (tr, truth), _ = archive
    mean = {1: [MEAN[1]], 2: MEAN, 3: MEAN + [0.3]}[len(channels)]
    std = {1: [STD[1]], 2: STD, 3: STD + [0.2]}[len(channels)]
    cfg = _configs(root, channels=channels, data_mean=mean, data_std=std, dem=True)
    ds = Dataset("train", cfg)
    zero_masks = 0
    for i in range(len(ds)):
        item = ds[i]
        t = truth[ds.records[i]["id"]]
        assert len(item) == 13
        for pos, key in ((2, "MS1"), (6, "SL1"), (9, "SL2")):
            want = _ref_normalize(_ref_concat(t[key][0], t[key][1], channels, 0.15), mean, std)
            assert item[pos].dtype == torch.float32 and torch.equal(item[pos], want), (i, key)
            assert torch.isfinite(item[pos]).all()
            assert item[pos - 2] == mean and item[pos - 1] == std
        assert item[3].dtype == torch.int64 and torch.equal(item[3], torch.from_numpy(t["mask"]).long())
        zero_masks += int(item[3].sum() == 0)
        dem = item[10]
        assert dem.shape == (1, 224, 224) and torch.isfinite(dem).all()
        filled = t["dem"].copy()
        filled[5:8, 9:12] = np.nan
        ok = ~np.isnan(filled)
        want_dem = (torch.from_numpy(t["dem"]) - 93.4313) / 1410.8382
        assert torch.equal(dem[0][torch.from_numpy(ok)], want_dem[torch.from_numpy(ok)])
        hole = dem[0, 5:8, 9:12] * 1410.8382 + 93.4313                       # filled from the nearest valid neighbours
        near = t["dem"][4:9, 8:13]
        assert float(hole.min()) >= np.nanmin(near) - 1e-2 and float(hole.max()) <= np.nanmax(near) + 1e-2
        assert (item[11], item[12]) == (t["clz"], t["act"])
    assert zero_masks >= 1                                                    # the cell without MK0_MLU: all zeros (Dataset.py:787-789)


def test_unscaled_minmax_and_target_range_modes(archive, tmp_path, monkeypatch):
    from kurosiwo_amd.dataset import Dataset
    root, (tr, truth), _ = archive
    monkeypatch.chdir(tmp_path)                                               # stats.pkl lands in the working directory, as in the reference
    ds = Dataset("train", _configs(root, scale_input=None, clamp_input=None))
    item = ds[0]
    t = truth[ds.records[0]["id"]]
    assert len(item) == 6 and torch.equal(item[0], _ref_concat(t["MS1"][0], t["MS1"][1], ["vv", "vh"], None))
    assert float(item[0].max()) == 200.0                                      # nan_to_num(image, 200) without a clamp (Dataset.py:167-168)
    ds = Dataset("train", _configs(root, scale_input="min-max"))
    item = ds[1]
    t = truth[ds.records[1]["id"]]
    act = ds.records[1]["activation"]
    assert os.path.exists("stats.pkl")
    ev = ds.min_max_random_events[act]
    lo = min(np.nanmin(np.where(truth[r["id"]]["valid"] == 1, truth[r["id"]]["MS1"][0], np.nan)) for r in ds.records if r["activation"] == act)
    assert ev["flood_vv"][0] == lo
    img = _ref_concat(t["MS1"][0], t["MS1"][1], ["vv", "vh"], 0.15)
    want = torch.cat([((img[c] - ev[f"flood_{n}"][0]) / (0.15 - ev[f"flood_{n}"][0]))[None] for c, n in enumerate(("vv", "vh"))])
    assert torch.equal(item[2], want) and item[1] == [0.15, 0.15]
    ds2 = Dataset("train", _configs(root, scale_input=[-1.0, 1.0]))
    assert torch.equal(ds2[1][2], torch.mul(want, torch.tensor(1.0) - torch.tensor(-1.0)) + torch.tensor(-1.0))
    with pytest.raises(NotImplementedError):
        Dataset("train", _configs(root, scale_input="custom"))[0]


def test_torch_dataloader_collates_the_reference_tuple(archive):
    from kurosiwo_amd.dataset import Dataset
    from kurosiwo_amd.synthetic import cd_inputs, seg_inputs
    root = archive[0]
    ds = Dataset("train", _configs(root, dem=True))
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=2)))
    assert len(batch) == 13 and batch[2].shape == (4, 2, 224, 224) and batch[3].shape == (4, 224, 224) and batch[10].shape == (4, 1, 224, 224)
    assert [v.shape for v in batch[0]] == [torch.Size([4])] * 2 and batch[12].tolist() == [ds.records[i]["activation"] for i in range(4)]
    (xa, xb), mask = cd_inputs(batch, ("pre_event_1", "post_event"), True)
    assert xa.shape == (4, 3, 224, 224) and torch.equal(xb[:, :2], batch[2])
    x, _ = seg_inputs(batch, ("pre_event_1", "pre_event_2", "post_event"), True)
    assert x.shape == (4, 7, 224, 224)


def test_batch_loader_raw_mode_and_rank_shards(archive):
    """TileBatchLoader(raw=True) needs no GPU: the raw tiles of a batch, one staging buffer, the rank's slice only"""
    from kurosiwo_amd.dataset import Dataset, ShardedBatch, TileBatchLoader
    from kurosiwo_amd.distributed import shard_batch
    root, (tr, truth), _ = archive
    ds = Dataset("train", _configs(root, dem=True))
    whole = list(TileBatchLoader(ds, 4, device="cpu", raw=True, threads=3))
    assert len(whole) == 2 and whole[0][2].shape == (4, 2, 224, 224) and whole[1][2].shape == (2, 2, 224, 224)
    assert len(list(TileBatchLoader(ds, 4, device="cpu", raw=True, drop_last=True))) == 1
    for b, batch in enumerate(whole):
        for j in range(batch[2].shape[0]):
            t = truth[ds.records[4 * b + j]["id"]]
            for pos, key in ((2, "MS1"), (6, "SL1"), (9, "SL2")):
                assert np.array_equal(batch[pos][j].numpy(), t[key], equal_nan=True)
            assert torch.equal(batch[3][j], torch.from_numpy(t["mask"]).long())
            assert torch.equal(batch[10][j], ds[4 * b + j][10])
            assert int(batch[11][j]) == t["clz"] and int(batch[12][j]) == t["act"]
    parts = [list(TileBatchLoader(ds, 4, device="cpu", raw=True, rank=r, world=2)) for r in (0, 1)]
    for b, batch in enumerate(whole):
        assert isinstance(parts[0][b], ShardedBatch) and shard_batch(parts[0][b], rank=0, world=2) is parts[0][b]
        for pos in (2, 3, 6, 9, 10, 11, 12):
            joined = torch.cat([parts[0][b][pos], parts[1][b][pos]])
            assert joined.shape == batch[pos].shape and np.array_equal(joined.numpy(), batch[pos].numpy(), equal_nan=True)
    a = [b[12].tolist() for b in TileBatchLoader(ds, 2, shuffle=True, device="cpu", raw=True, seed=5)]
    b_ = [b[12].tolist() for b in TileBatchLoader(ds, 2, shuffle=True, device="cpu", raw=True, seed=5)]
    assert a == b_ and sorted(sum(a, [])) == sorted(r["activation"] for r in ds.records)
    with pytest.raises(ValueError):
        TileBatchLoader(ds, 3, device="cpu", raw=True, world=2)
    with pytest.raises(RuntimeError):
        next(iter(TileBatchLoader(ds, 2, device="cpu")))                      # normalising is GPU work: no CPU fallback


def test_slc_dataset_with_ragged_tiles(tmp_path):
    from make_synthetic_archive import make
    from kurosiwo_amd.dataset import SLCDataset
    root = str(tmp_path)
    grids, truth = make(root, TRAIN, tiles_per_act=3, seed=4, slc=True, ragged=True)
    os.makedirs(os.path.join(root, "pickle"))
    json.dump(grids, open(os.path.join(root, "pickle", "train.json"), "w"))
    slc_mean, slc_std = [0.022367, 39.242, 81.13, 0.043526], [1.2843, 25.6152, 58.0151, 1.2844]
    cfg = _configs(root, slc=True, slc_root_path=root, train_json=os.path.join(root, "pickle", "train.json"),
                   test_json=os.path.join(root, "pickle", "train.json"), slc_mean=slc_mean, slc_std=slc_std, dem=True,
                   slc_dem_mean=[93.4313], slc_dem_std=[1410.8382])
    ds = SLCDataset("train", cfg)
    assert len(ds) == 6
    padded = 0
    for i in range(len(ds)):
        item = ds[i]
        t = truth[ds.records[i]["id"]]
        assert len(item) == 13 and item[2].shape == (4, 224, 224) and item[3].shape == (224, 224)
        h, w = t["MS1"].shape[1:]
        top, left = (224 - h) // 2, (224 - w) // 2
        fill = int(t["MS1"].mean())
        for pos, key in ((2, "MS1"), (6, "SL1"), (9, "SL2")):
            want = np.full((4, 224, 224), fill, np.float32)
            want[:, top:top + h, left:left + w] = t[key]
            assert torch.equal(item[pos], _ref_normalize(torch.from_numpy(want), slc_mean, slc_std))
        m = np.full((224, 224), 3, np.int64)
        m[top:top + h, left:left + w] = t["mask"]
        assert torch.equal(item[3], torch.from_numpy(m))
        assert torch.isfinite(item[10]).all()                                   # the 3.4e38 no-data cells were filled
        padded += int((h, w) != (224, 224))
    assert padded == 2


def test_slc_batch_loader_raw_mode_and_ragged_fallback(tmp_path):
    """the 4-band batch path (ksmi_tile_batch_read_bands): raw tiles of a batch equal what was written; a batch holding a ragged tile
    falls back to the per-sample path (the reference pads such tiles, dataset/Dataset.py:1173-1207) and yields the same tuple"""
    from make_synthetic_archive import make
    from kurosiwo_amd.dataset import SLCDataset, TileBatchLoader
    root = str(tmp_path)
    grids, truth = make(root, TRAIN, tiles_per_act=3, seed=4, slc=True, ragged=True)
    os.makedirs(os.path.join(root, "pickle"))
    json.dump(grids, open(os.path.join(root, "pickle", "train.json"), "w"))
    slc_mean, slc_std = [0.022367, 39.242, 81.13, 0.043526], [1.2843, 25.6152, 58.0151, 1.2844]
    cfg = _configs(root, slc=True, slc_root_path=root, train_json=os.path.join(root, "pickle", "train.json"),
                   test_json=os.path.join(root, "pickle", "train.json"), slc_mean=slc_mean, slc_std=slc_std, dem=False)
    ds = SLCDataset("train", cfg)
    sizes = [truth[r["id"]]["MS1"].shape[1:] for r in ds.records]
    assert sizes.count((224, 224)) == 4 and len(sizes) == 6
    full = [i for i, s in enumerate(sizes) if s == (224, 224)]
    ld = TileBatchLoader(ds, 2, device="cpu", raw=True, prefetch=0)
    b = ld.load(full[:2])
    for j, i in enumerate(full[:2]):
        t = truth[ds.records[i]["id"]]
        for pos, key in ((2, "MS1"), (6, "SL1"), (9, "SL2")):
            assert np.array_equal(b[pos][j].numpy(), t[key])
        assert torch.equal(b[3][j], torch.from_numpy(t["mask"]).long())
    ragged = [i for i, s in enumerate(sizes) if s != (224, 224)][0]
    b = ld.load([full[0], ragged])                                           # -> per-sample path: normalised, padded
    want = torch.utils.data.default_collate([ds[full[0]], ds[ragged]])
    assert len(b) == len(want) == 12
    for pos in (2, 3, 6, 9):
        assert torch.equal(b[pos], want[pos])
