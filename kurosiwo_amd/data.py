"""Loaders.  `prepare_loaders` keeps the reference signature (utilities/utilities.py:73-126).  With the Kuro Siwo archive on disk
(configs["root_path"]/data + the grid pickles, or configs["slc_root_path"] + the json indices) it serves the archive through
kurosiwo_amd/dataset.py (native GeoTIFF reader; GRD + "normalize": the batch-level TileBatchLoader with the GPU-side preprocess);
without it, synthetic tiles with the exact collated-batch layout (kurosiwo_amd/synthetic.py; there is no archive in the build
image)."""
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
    archive = archive_kind(configs)
    if archive is not None:
        return _archive_loaders(configs, archive, bs, workers)
    from . import distributed as D
    ds = {m: SyntheticCDDataset(m, configs) for m in ("train", "val", "test")}
    # (world > 1: every rank loads only its slice of each global batch, distributed.RankShardBatchSampler)
    mk = lambda m, shuffle, drop: D.make_loader(ds[m], bs, shuffle, drop, workers, seed=configs.get("seed", 999))
    tr, va, te = mk("train", True, True), mk("val", False, False), mk("test", False, False)
    print("Samples in Train Set: ", len(ds["train"]))
    print("Samples in Val Set: ", len(ds["val"]))
    print("Samples in Test Set: ", len(ds["test"]))
    return tr, va, te


def archive_kind(configs):
    """"slc" / "grd" when the archive the configs point to is on disk, else None (-> synthetic tiles); KSMI_DATA=synthetic forces
    None, KSMI_DATA=archive makes a missing archive an error instead of a silent switch"""
    import os
    want = os.environ.get("KSMI_DATA", "")
    if want == "synthetic":
        return None
    kind = None
    if configs.get("slc"):
        if configs.get("slc_root_path") and os.path.isdir(configs["slc_root_path"]) and os.path.isfile(str(configs.get("train_json"))):
            kind = "slc"
    elif configs.get("root_path") and os.path.isdir(os.path.join(configs["root_path"], "data")) and os.path.isfile(str(configs.get("train_pickle"))):
        kind = "grd"
    if kind is None and want == "archive":
        raise FileNotFoundError("KSMI_DATA=archive but the archive / grid index of the configs is not on disk")
    return kind


def _loader_threads(configs, world):
    """decode threads of a rank's TileBatchLoader: configs["loader_threads"], else 32 (measured on the 256-core MI355X host,
    profiles/r03_loader_probe.jsonl: 4.3 k grid cells/s per process) capped at this rank's share of the host cores"""
    import os
    want = configs.get("loader_threads")
    return int(want) if want else max(1, min(32, (os.cpu_count() or 8) // max(1, world)))


def _archive_loaders(configs, kind, bs, workers):
    """utilities/utilities.py:87-121 on the archive"""
    from . import dataset as DS
    from . import distributed as D
    print("=" * 20)
    print("Initializing ", configs["track"])
    print("=" * 20)
    cls = DS.SLCDataset if kind == "slc" else DS.Dataset
    ds = {m: cls(mode=m, configs=configs) for m in ("train", "val", "test")}
    batch_level = ((kind == "slc" or configs.get("clamp_input") is not None) and configs.get("scale_input") == "normalize"
                   and not configs.get("uint8") and not configs.get("slope") and not configs.get("oversampling")
                   and configs.get("gpu_input_pipeline", True) and str(configs.get("device", "cuda")).startswith("cuda")
                   and torch.cuda.is_available())
    if batch_level:
        mk = lambda m, shuffle, drop: DS.TileBatchLoader(ds[m], bs, shuffle=shuffle, drop_last=drop, device=configs.get("device", "cuda"),
                                                         threads=_loader_threads(configs, D.world_size()), rank=D.get_rank(), world=D.world_size(),
                                                         seed=configs.get("seed", 999) if shuffle else None)
    else:
        mk = lambda m, shuffle, drop: D.make_loader(ds[m], bs, shuffle, drop, workers, seed=configs.get("seed", 999))
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
