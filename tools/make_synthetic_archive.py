"""A small archive with the on-disk layout of Kuro Siwo (GRD: <root>/data/<actid>/<aoi>/<hash>/{MS1_IVV,MS1_IVH,SL1_*,SL2_*,MK0_MLU,
MK0_MNA,MK0_DEM}_*.tif + a gzip grid pickle; SLC: 4-band MS1/SL1/SL2 tiles + a json index) filled with the synthetic tiles of
kurosiwo_amd/synthetic.py, for tests and for running main.py end to end through the archive path (there is no network and no
archive in the build image).

    python tools/make_synthetic_archive.py /tmp/ks --tiles 48           # then: root_path=/tmp/ks in the data config
"""
import argparse
import gzip
import json
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def make(root, acts, tiles_per_act=4, seed=7, slc=False, dem=True, ragged=False):
    """returns (grid dict, {record id: raw arrays}); raw SAR tiles carry NaN no-data, negatives and values above the clamp"""
    import torch
    from kurosiwo_amd import geotiff as G
    from kurosiwo_amd.synthetic import make_batch, DATA_MEAN, DATA_STD
    rng = np.random.default_rng(seed)
    grids, truth = {}, {}
    codecs = [dict(compression="lzw", predictor=3), dict(compression="deflate", predictor=3), dict(compression=None), dict(compression="deflate", tile=(128, 128))]
    n = 0
    for act in acts:
        for t in range(tiles_per_act):
            aoi = 1 + t % 2
            rel = os.path.join(str(act), f"{aoi:02}", f"{n:08x}")
            folder = os.path.join(root, "data" if not slc else "", rel)
            os.makedirs(folder, exist_ok=True)
            b = make_batch(1, seed=int(rng.integers(1 << 30)), dem=True, channels=4 if slc else 2)
            mean = np.array((DATA_MEAN * 2)[:b[2].shape[1]], np.float32)[:, None, None]
            std = np.array((DATA_STD * 2)[:b[2].shape[1]], np.float32)[:, None, None]
            raw = {k: (b[i][0].numpy() * std + mean).astype(np.float32) for k, i in (("MS1", 2), ("SL1", 6), ("SL2", 9))}
            for k in raw:                                  # what the Dataset's clamp / nan_to_num exists for
                u = rng.random(raw[k].shape)
                if not slc:                                # (the SLC class has no NaN handling: dataset/Dataset.py:1173-1181 takes int(mean))
                    raw[k][u < 0.01] = np.nan
                raw[k][(u > 0.01) & (u < 0.015)] = -0.003
                raw[k][(u > 0.015) & (u < 0.02)] = 0.9
            mask = b[3][0].numpy().astype(np.uint8)
            valid = ((mask != 3) & ~np.any([np.isnan(v).any(0) for v in raw.values()], axis=0)).astype(np.uint8)    # no-data is never "valid"
            demv = (b[10][0, 0].numpy() * 1410.8382 + 93.4313).astype(np.float32)
            demv[5:8, 9:12] = np.nan
            kw = codecs[n % len(codecs)]
            geo = dict(pixel_scale=(10.0, 10.0), origin=(500000.0 + 2240 * n, 4.2e6))
            stamp = f"{act}_{aoi:02}_2021010{1 + t % 9}"
            if slc:
                hw = (200, 216) if ragged and n % 3 == 0 else (224, 224)
                for k in raw:
                    raw[k] = raw[k][:, :hw[0], :hw[1]]
                    G.write(os.path.join(folder, f"{k}_{stamp}.tif"), raw[k], **kw, **geo)
                mask, valid = mask[:hw[0], :hw[1]], valid[:hw[0], :hw[1]]
                demn = np.where(np.isnan(demv), np.float32(3.4e38), demv)[:hw[0], :hw[1]]
                G.write(os.path.join(folder, f"MK0_DEM_{stamp}.tif"), demn, nodata=3.4e38, compression="deflate", predictor=3, **geo)
            else:
                for k in raw:
                    G.write(os.path.join(folder, f"{k}_IVV_{stamp}.tif"), raw[k][0], nodata=float("nan"), **kw, **geo)
                    G.write(os.path.join(folder, f"{k}_IVH_{stamp}.tif"), raw[k][1], nodata=float("nan"), **kw, **geo)
                if dem:
                    G.write(os.path.join(folder, f"MK0_DEM_{stamp}.tif"), demv, nodata=float("nan"), compression="lzw", predictor=3, **geo)
                open(os.path.join(folder, f"MS1_IVV_{stamp}.tif.aux.xml"), "w").write("<PAMDataset/>")       # sidecars are skipped
            if not (not slc and n % 5 == 4):                                                             # some cells have no label mask
                G.write(os.path.join(folder, f"MK0_MLU_{stamp}.tif"), mask, compression="deflate", predictor=2)
            else:
                mask = np.zeros_like(mask)
            G.write(os.path.join(folder, f"MK0_MNA_{stamp}.tif"), valid, compression="packbits")
            clz = int(1 + n % 3)
            key = f"{n:08x}"
            info = {"actid": int(act), "aoiid": aoi}
            grids[key] = dict(path=rel, clz=clz, **info) if slc else dict(path=rel, clz=clz, info=info)
            truth[key] = dict(raw, mask=mask, valid=valid, dem=demv, clz=clz, act=int(act))
            n += 1
    return grids, truth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--tiles", type=int, default=48, help="tiles per split")
    ap.add_argument("--slc", action="store_true")
    a = ap.parse_args()
    from kurosiwo_amd.config import load_json5
    cfg = load_json5(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "configs", "train", "data_config.json"))
    acts = {"train": cfg["train_acts"][:4], "val": cfg["val_acts"][:2], "test": cfg["test_acts"][:2]}
    os.makedirs(os.path.join(a.root, "pickle"), exist_ok=True)
    tr, _ = make(a.root, acts["train"], max(1, a.tiles // 4), seed=1, slc=a.slc)
    te, _ = make(a.root, acts["val"] + acts["test"], max(1, a.tiles // 2), seed=2, slc=a.slc)
    if a.slc:
        json.dump(tr, open(os.path.join(a.root, "pickle", "train.json"), "w"))
        json.dump(te, open(os.path.join(a.root, "pickle", "test.json"), "w"))
    else:
        pickle.dump(tr, gzip.open(os.path.join(a.root, "pickle", "train.gz"), "wb"))
        pickle.dump(te, gzip.open(os.path.join(a.root, "pickle", "test.gz"), "wb"))
    print(f"{len(tr)} train cells, {len(te)} val+test cells under {a.root}")


if __name__ == "__main__":
    main()
