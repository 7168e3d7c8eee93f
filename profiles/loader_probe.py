"""N4 measurement: grid cells per second from an archive on local disk (page cache warm) into normalised device tensors:
(a) per-sample Dataset + torch DataLoader with k worker processes (the reference's arrangement, utilities/utilities.py:96-121, with
the native tile reader in place of cv2), (b) dataset.TileBatchLoader with k decode threads (one staging buffer, one copy, GPU
preprocess), (c) the same handing out raw tiles (SNUNet's fused first convolution).
    python profiles/loader_probe.py [cells]   -> one json line per arrangement"""
import gzip
import json
import os
import pickle
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from make_synthetic_archive import make
    from test_dataset_cpu import TRAIN, _configs
    from kurosiwo_amd.dataset import Dataset, TileBatchLoader
    cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    root = tempfile.mkdtemp(prefix="ks_probe_")
    os.makedirs(os.path.join(root, "pickle"))
    tr, _ = make(root, TRAIN, tiles_per_act=cells // 2, seed=1)
    pickle.dump(tr, gzip.open(os.path.join(root, "pickle", "train.gz"), "wb"))
    pickle.dump({}, gzip.open(os.path.join(root, "pickle", "test.gz"), "wb"))
    cfg = _configs(root, dem=True, device="cuda")
    ds = Dataset("train", cfg)
    B = 32
    print(json.dumps({"cells": len(ds), "batch": B, "host_cores": os.cpu_count(), "torch_threads": torch.get_num_threads()}))

    def timed(name, it_fn, **kw):
        for _ in it_fn():          # warm: page cache, worker start-up
            pass
        torch.cuda.synchronize()
        t = time.time()
        n = 0
        for rep in range(2):
            for b in it_fn():
                x = b[2].to("cuda", non_blocking=True)
                n += x.shape[0]
        torch.cuda.synchronize()
        dt = time.time() - t
        print(json.dumps(dict(arrangement=name, cells_per_s=round(n / dt, 1), **kw)), flush=True)
    for w in (0, 8, 16):
        timed("per-sample Dataset + torch DataLoader", lambda: iter(torch.utils.data.DataLoader(ds, batch_size=B, num_workers=w, pin_memory=True)), workers=w)
    for th in (1, 4, 8, 16, 32, 64):
        timed("TileBatchLoader (normalised on the GPU)", lambda: iter(TileBatchLoader(ds, B, device="cuda", threads=th)), threads=th)
    timed("TileBatchLoader, no prefetch thread", lambda: iter(TileBatchLoader(ds, B, device="cuda", threads=32, prefetch=0)), threads=32)
    torch.set_num_threads(8)
    timed("TileBatchLoader, no prefetch thread, torch.set_num_threads(8)", lambda: iter(TileBatchLoader(ds, B, device="cuda", threads=32, prefetch=0)), threads=32)
    timed("per-sample Dataset + torch DataLoader, torch.set_num_threads(8)", lambda: iter(torch.utils.data.DataLoader(ds, batch_size=B, num_workers=0, pin_memory=True)), workers=0)
    timed("TileBatchLoader raw (normalised in the first conv)", lambda: iter(TileBatchLoader(ds, B, device="cuda", threads=32, raw=True)), threads=32)


if __name__ == "__main__":
    main()
