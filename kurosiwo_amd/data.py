"""Loaders.  The reference's Dataset classes (dataset/Dataset.py) need the Kuro Siwo archive,
cv2/rioxarray and grid pickles that are not part of this build (out of scope);
`prepare_loaders` keeps the reference signature (utilities/utilities.py:73-126) and serves
synthetic tiles with the exact collated-batch layout (kurosiwo_amd/synthetic.py)."""
import torch

from .synthetic import make_batch


class SyntheticCDDataset(torch.utils.data.Dataset):
    def __init__(self, mode, configs):
        import os
        n = configs.get("synthetic_tiles", {"train": 256, "val": 64, "test": 64})[mode]
        if os.environ.get("KSMI_SYNTHETIC_TILES"):          # "train,val,test" override (tests)
            n = int(os.environ["KSMI_SYNTHETIC_TILES"].split(",")[("train", "val", "test").index(mode)])
        self.n, self.mode, self.cfg = n, mode, configs
        self.activations = list(configs[{"train": "train_acts", "val": "val_acts", "test": "test_acts"}[mode]])
        self.seed0 = {"train": 999, "val": 424242, "test": 515151}[mode]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        # SLC tiles: 4 bands per date (utilities/utilities.py:386-390 doubles the SAR channels)
        nch = len(self.cfg["channels"]) * (2 if self.cfg.get("slc") else 1)
        b = make_batch(1, seed=self.seed0 + i, dem=bool(self.cfg.get("dem")), channels=nch)
        out = []
        for t in b:
            if isinstance(t, list):
                out.append([float(x[0]) for x in t])
            else:
                out.append(t[0])
        out[-1] = torch.tensor(self.activations[i % len(self.activations)], dtype=torch.int64)
        return tuple(out)


def prepare_loaders(configs):
    if configs["track"] not in ["RandomEvents"]:
        print("No such track! We currently support only RandomEvents")
        raise SystemExit(2)
    bs, workers = configs["batch_size"], configs.get("num_workers", 0)
    ds = {m: SyntheticCDDataset(m, configs) for m in ("train", "val", "test")}
    mk = lambda m, shuffle, drop: torch.utils.data.DataLoader(ds[m], batch_size=bs, shuffle=shuffle, num_workers=workers,
                                                              pin_memory=True, drop_last=drop)
    tr, va, te = mk("train", True, True), mk("val", False, False), mk("test", False, False)
    print("Samples in Train Set: ", len(ds["train"]))
    print("Samples in Val Set: ", len(ds["val"]))
    print("Samples in Test Set: ", len(ds["test"]))
    return tr, va, te


def preprocess_gpu(raw, mean, std, clamp_input=0.15, out=None):
    """The Dataset's clamp -> nan_to_num -> Normalize (dataset/Dataset.py:164-168,193-198) on a raw [B,C,H,W] fp32 CUDA tensor."""
    import torch
    from . import _lib
    from .runtime import require_gpu, stream_ptr
    require_gpu(raw)
    raw = raw.contiguous().float()
    B, Cc, H, W = raw.shape
    m = torch.as_tensor(mean, dtype=torch.float32, device=raw.device)
    s = torch.as_tensor(std, dtype=torch.float32, device=raw.device)
    out = torch.empty_like(raw) if out is None else out
    _lib.check(_lib.load().ksmi_sar_preprocess(raw.data_ptr(), m.data_ptr(), s.data_ptr(), out.data_ptr(), B, Cc, H * W, float(clamp_input), stream_ptr()),
               "sar_preprocess")
    return out
