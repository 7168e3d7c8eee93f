"""The Kuro Siwo archive on disk: the reference's Dataset classes (dataset/Dataset.py) over the native GeoTIFF
tile reader (kurosiwo_amd/geotiff.py -> ksmi_tiff_* / ksmi_tile_batch_read), plus a batch-level loader that replaces the
"8 DataLoader workers + cv2 + per-sample torch ops" pipeline (utilities/utilities.py:96-121) by

    thread pool decode of the whole batch -> ONE pinned staging buffer -> ONE host-to-device copy -> clamp / nan_to_num / Normalize
    on the GPU (ksmi_sar_preprocess, or inside SNUNet's first convolution: model.set_input_pipeline)

`Dataset`, `SLCDataset` keep the reference's constructor, record selection, file naming (MS1_IVV/IVH, SL1_*, SL2_*, MK0_MLU,
MK0_MNA, MK0_DEM) and the tuple `__getitem__` returns (dataset/Dataset.py:824-860), so torch's DataLoader works on them exactly as
in the reference; `TileBatchLoader` yields the same tuple, collated, with the images already on the device.

Not carried over (need packages that are not in the image): albumentations views (`data_augmentations`, task
"self-supervised" of Dataset.create_views), scale_input == "custom" (torchio RescaleIntensity), the "diffusion-unsup" per-date
records.  They raise NotImplementedError instead of silently doing something else."""
import bz2
import gzip
import lzma
import os
import pickle
import random

import numpy as np
import torch

from . import geotiff

TILE = 224
SAR_PREFIXES = ("MS1_IVV", "MS1_IVH", "SL1_IVV", "SL1_IVH", "SL2_IVV", "SL2_IVH")


def get_grids(pickle_path):
    """dataset/Dataset.py:27-33 (compress_pickle.load: the codec follows the file, .gz in the published configs)"""
    if not os.path.isfile(pickle_path):
        print("Pickle file not found! ", pickle_path)
        raise SystemExit(2)
    with open(pickle_path, "rb") as f:
        magic = f.read(6)
    opener = gzip.open if magic[:2] == b"\x1f\x8b" else bz2.open if magic[:3] == b"BZh" else lzma.open if magic == b"\xfd7zXZ\x00" else open
    with opener(pickle_path, "rb") as f:
        return pickle.load(f)


def _tile_files(folder):
    """prefix -> path for the tiles of one grid cell (the `for file in files` chains of Dataset.__getitem__, :659-730: every
    non-xml file is tried against the prefixes; a later file with the same prefix wins, as there)"""
    out = {}
    for name in os.listdir(folder):
        if "xml" in name:
            continue
        for prefix in SAR_PREFIXES + ("MK0_MLU", "MK0_MNA", "MK0_DEM", "MS1", "SL1", "SL2"):
            if name.startswith(prefix):
                out[prefix] = os.path.join(folder, name)
    return out


def fill_nodata_nearest(a):
    """rioxarray's DataArray.rio.interpolate_na() with its default method "nearest" (dataset/Dataset.py:733-735): every NaN takes the
    value of the nearest valid pixel (Euclidean, pixel grid; rioxarray searches in model coordinates with scipy's griddata, the same
    thing for the square pixels of the archive; equidistant ties may resolve differently).  Native: ksmi_tiles_fill_nodata."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    a = a.copy() if not a.flags["WRITEABLE"] or a.base is not None else a
    return geotiff.fill_nodata(a)


def slope_riserun(dem, nodata=None, cell=1.0):
    """richdem.TerrainAttribute(rdarray(dem), attrib="slope_riserun") (dataset/Dataset.py:748-752): Horn's 3x3 finite differences,
    neighbours outside the grid or equal to no_data replaced by the centre cell, cell size 1 (an rdarray made from a bare numpy
    array carries no geotransform), no_data centres stay no_data.  Restated from richdem's documentation; richdem is not in this
    image, so this function is unpinned."""
    z = np.asarray(dem, dtype=np.float64)
    H, W = z.shape
    pad = np.full((H + 2, W + 2), np.nan)
    pad[1:-1, 1:-1] = z
    if nodata is not None and nodata == nodata:
        pad[pad == nodata] = np.nan

    def nb(dy, dx):
        v = pad[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
        return np.where(np.isnan(v), z, v)
    a, b, c = nb(-1, -1), nb(-1, 0), nb(-1, 1)
    d, f = nb(0, -1), nb(0, 1)
    g, h, i = nb(1, -1), nb(1, 0), nb(1, 1)
    dzdx = ((c + 2 * f + i) - (a + 2 * d + g)) / 8.0 / cell
    dzdy = ((g + 2 * h + i) - (a + 2 * b + c)) / 8.0 / cell
    out = np.sqrt(dzdx * dzdx + dzdy * dzdy)
    if nodata is not None and nodata == nodata:
        out[z == nodata] = nodata
    return out.astype(np.float32)


class _Records:
    """record selection shared by the GRD and SLC classes (dataset/Dataset.py:37-146, 986-1078)"""

    def _select(self, total_grids, configs, info_of):
        all_activations = list(self.train_acts) + list(self.val_acts) + list(self.test_acts)
        self.records, self.positive_records, self.negative_records = [], [], []
        for key in total_grids:
            info = info_of(total_grids[key])
            record = {"id": key, "path": total_grids[key]["path"], "info": info, "type": None, "clz": total_grids[key]["clz"]}
            activation, aoi = info["actid"], info["aoiid"]
            act_aoi = str(activation) + "_" + f"{aoi:02}" if configs["track"] == "Climatic" else activation
            record["activation"] = activation
            if act_aoi in self.valid_acts:
                self.clz_stats[record["clz"]] += 1
                self.act_stats[act_aoi] = self.act_stats.get(act_aoi, 0) + 1
                self.records.append(record)
                (self.positive_records if key in self.grids else self.negative_records).append(record)
            if act_aoi not in all_activations and act_aoi not in self.non_valids:
                print("Activation: ", activation, " not in Activations")
                self.non_valids.append(act_aoi)
        print("Samples per Climatic zone for mode: ", self.mode)
        print(self.clz_stats)
        print("Samples per Activation for mode: ", self.mode)
        print(self.act_stats)
        self.num_examples = len(self.records)
        self.activations = set(r["activation"] for r in self.records)

    def _init_common(self, mode, configs):
        if configs["task"] in ("self-supervised", "diffusion-unsup") or configs.get("data_augmentations"):
            raise NotImplementedError("albumentations views / per-date diffusion records are outside this build (SURVEY.md §2)")
        self.train_acts, self.val_acts, self.test_acts = configs["train_acts"], configs["val_acts"], configs["test_acts"]
        self.mode, self.configs = mode, configs
        self.augmentations = None
        self.non_valids = []
        self.clz_stats, self.act_stats = {1: 0, 2: 0, 3: 0}, {}
        self.valid_acts = {"train": self.train_acts, "val": self.val_acts}.get(mode, self.test_acts)

    def __len__(self):
        return self.num_examples

    def _sample(self, index):
        if self.configs.get("oversampling") and self.mode == "train":            # dataset/Dataset.py:641-649
            pool = self.positive_records if random.randint(0, 1) == 0 else self.negative_records
            return pool[random.randint(0, len(pool) - 1)]
        return self.records[index]


class Dataset(_Records, torch.utils.data.Dataset):
    """GRD tiles: dataset/Dataset.py:36-860."""

    def __init__(self, mode="train", configs=None):
        self._init_common(mode, configs)
        self.root_path = os.path.join(configs["root_path"], "data")
        self.pickle_path = configs["train_pickle"] if mode == "train" else configs["test_pickle"]
        self._min_max = None
        self.negative_grids = None
        self.grids = get_grids(self.pickle_path)
        total = dict(self.grids)
        if configs.get("oversampling") and mode == "train":
            self.negative_grids = get_grids(configs["negative_pickle"])
            total.update(self.negative_grids)
            print("=" * 20)
            print("Enabling oversampling")
            print("Length of positive grids: ", len(self.grids))
            print("Length of negative grids: ", len(self.negative_grids))
            print("Total grids: ", len(total))
            print("=" * 20)
        self._select(total, configs, lambda g: g["info"])

    # ---- per-date image assembly -----------------------------------------------------------------
    def channel_stack(self, vv, vh):
        """the np.vstack of Dataset.concat (:148-162), before clamp and nan_to_num"""
        ch = self.configs["channels"]
        if set(ch) == {"vv", "vh", "vh/vv"}:
            return np.stack((vv, vh, vh / (vv + 1e-7)))
        if set(ch) == {"vv", "vh"}:
            return np.stack((vv, vh))
        if ch == ["vh"]:
            return vh[None]
        raise ValueError(f"unsupported channels {ch}")

    def concat(self, image1, image2):
        """dataset/Dataset.py:148-169"""
        image = torch.from_numpy(self.channel_stack(image1, image2)).float()
        if self.configs["clamp_input"] is not None:
            image = torch.clamp(image, min=0.0, max=self.configs["clamp_input"])
            return torch.nan_to_num(image, self.configs["clamp_input"])
        return torch.nan_to_num(image, 200)

    @property
    def min_max_random_events(self):
        """dataset/Dataset.py:498-638 (computed when a scaling mode first asks for it; the reference computes it in __init__)"""
        if self._min_max is None:
            self._min_max = self.update_min_max_stats()
        return self._min_max

    def update_min_max_stats(self):
        if os.path.exists("stats.pkl"):
            print(f"({self.mode}) Using precalculated stats for dataset...")
            return get_grids("stats.pkl")
        print(f"({self.mode}) Calculating stats for dataset...")
        stats = {}
        names = {"pre1": "SL1", "pre2": "SL2", "flood": "MS1"}
        for mode in ("train", "val", "test"):
            valid = self.configs[f"{mode}_acts"]
            grids = get_grids(self.configs["train_pickle"] if mode == "train" else self.configs["test_pickle"])
            for key in grids:
                info = grids[key]["info"]
                act = info["actid"]
                act_aoi = str(act) + "_" + f"{info['aoiid']:02}" if self.configs["track"] == "Climatic" else act
                if act_aoi in self.non_valids or act_aoi not in valid:
                    continue
                files = _tile_files(os.path.join(self.root_path, grids[key]["path"]))
                ok = geotiff.read(files["MK0_MNA"])[0] == 1
                cur = stats.setdefault(act, {})
                for img, pre in names.items():
                    for pol in ("vv", "vh"):
                        v = geotiff.read(files[f"{pre}_I{pol.upper()}"], dtype=np.float32)[0][ok]
                        lo, hi = (v.min(), v.max()) if v.size else (np.inf, -np.inf)
                        k = f"{img}_{pol}"
                        cur[k] = (min(lo, cur[k][0]), max(hi, cur[k][1])) if k in cur else (lo, hi)
        print(f"({self.mode}) New stats:")
        print(stats)
        with open("stats.pkl", "wb") as f:
            pickle.dump(stats, f)
        return stats

    def _min_max_of(self, img_name, activation):
        ev, ch, clamp = self.min_max_random_events[activation], self.configs["channels"], self.configs["clamp_input"]
        mins, maxs = {}, {}
        for c in ("vv", "vh"):
            if c in ch:
                mins[c] = ev[f"{img_name}_{c}"][0]
                maxs[c] = clamp if clamp is not None else ev[f"{img_name}_{c}"][1]
        if "vh/vv" in ch:
            mins["vh/vv"] = ev[f"{img_name}_vh"][0] / ev[f"{img_name}_vv"][0]
            maxs["vh/vv"] = 1.0 if clamp is not None else ev[f"{img_name}_vh"][1] / ev[f"{img_name}_vv"][1]
        return mins, maxs

    def scale_img(self, img, valid_mask, img_name, activation):
        """dataset/Dataset.py:192-333: "normalize", "min-max", [new_min, new_max]"""
        mode = self.configs["scale_input"]
        if mode == "normalize":
            means, stds = self.configs["data_mean"], self.configs["data_std"]
            m = torch.as_tensor(means, dtype=img.dtype).view(-1, 1, 1)
            s = torch.as_tensor(stds, dtype=img.dtype).view(-1, 1, 1)
            return means, stds, (img - m) / s                                   # torchvision.transforms.Normalize: sub_ then div_
        if mode == "min-max" or isinstance(mode, list):
            mins, maxs = self._min_max_of(img_name, activation)
            ch = self.configs["channels"]
            new = torch.cat([((img[i] - mins[c]) / (maxs[c] - mins[c]))[None] for i, c in enumerate(ch)], dim=0)
            if isinstance(mode, list):
                new_min, new_max = (torch.tensor(v) for v in mode)
                new = torch.mul(new, (new_max - new_min)) + new_min
            return list(mins.values()), list(maxs.values()), new
        raise NotImplementedError(f"scale_input = {mode!r} (torchio's RescaleIntensity is not in this image)")

    # ---- one grid cell ------------------------------------------------------------------------------
    def _read(self, path):
        a = geotiff.read(path, dtype=np.float32)[0]                               # cv.imread(path, cv.IMREAD_ANYDEPTH) on a float32 tile
        if self.configs.get("uint8"):                                             # dataset/Dataset.py:672-675
            a = a / a.max()
            a = (a * 255).astype(np.uint8)
        return a

    def read_dem(self, path):
        """dataset/Dataset.py:728-779: the DEM (gaps filled from the nearest valid pixel) or its slope, standardised"""
        cfg = self.configs
        a, meta = geotiff.read(path, dtype=np.float32, squeeze=False)
        a = np.stack([fill_nodata_nearest(b) for b in a])
        if not cfg["dem"] and cfg.get("slope"):
            print("To return the slope the DEM option must be enabled. Validate the config file!")
            raise SystemExit(2)
        if cfg.get("slope"):
            out = torch.from_numpy(slope_riserun(a[0], meta["nodata"])[None])
            mean, std = cfg["slope_mean"], cfg["slope_std"]
        else:
            out = torch.from_numpy(a)
            mean, std = cfg["dem_mean"], cfg["dem_std"]
        if cfg["scale_input"] is not None:
            out = (out - torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)) / torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)
        return out

    def sample_files(self, index):
        sample = self._sample(index)
        return sample, _tile_files(os.path.join(self.root_path, sample["path"]))

    def __getitem__(self, index):
        sample, files = self.sample_files(index)
        cfg = self.configs
        mask = geotiff.read(files["MK0_MLU"])[0] if "MK0_MLU" in files else np.zeros((TILE, TILE))
        valid_mask = torch.from_numpy(geotiff.read(files["MK0_MNA"])[0])
        flood = self.concat(self._read(files["MS1_IVV"]), self._read(files["MS1_IVH"]))
        pre_event_1 = self.concat(self._read(files["SL1_IVV"]), self._read(files["SL1_IVH"]))
        pre_event_2 = self.concat(self._read(files["SL2_IVV"]), self._read(files["SL2_IVH"]))
        dem = self.read_dem(files["MK0_DEM"]) if "MK0_DEM" in files and cfg["dem"] else None
        mask = torch.from_numpy(np.asarray(mask)).long()
        clz, activation = sample["clz"], sample["activation"]
        if cfg["scale_input"] is None:
            return (flood, mask, pre_event_1, pre_event_2) + ((dem,) if cfg["dem"] else ()) + (clz, activation)
        valid_mask = valid_mask == 1
        f1, f2, flood = self.scale_img(flood, valid_mask, "flood", activation)
        p11, p12, pre_event_1 = self.scale_img(pre_event_1, valid_mask, "pre1", activation)
        p21, p22, pre_event_2 = self.scale_img(pre_event_2, valid_mask, "pre2", activation)
        return (f1, f2, flood, mask, p11, p12, pre_event_1, p21, p22, pre_event_2) + ((dem,) if cfg["dem"] else ()) + (clz, activation)


class SLCDataset(_Records, torch.utils.data.Dataset):
    """4-band SLC tiles: dataset/Dataset.py:986-1228 (one multi-band GeoTIFF per date, a json grid index)."""

    def __init__(self, mode="train", configs=None):
        from .config import load_json5
        print("=" * 20)
        print("Initializing SLC Dataset")
        print("=" * 20)
        self._init_common(mode, configs)
        self.root_path = os.path.join(configs["slc_root_path"])
        self.pickle_path = configs["train_json"] if mode == "train" else configs["test_json"]
        self.negative_grids = None
        self.grids = load_json5(self.pickle_path)
        self._select(self.grids, configs, lambda g: g)

    def normalize(self, image):
        means, stds = self.configs["slc_mean"], self.configs["slc_std"]
        m = torch.as_tensor(means, dtype=image.dtype).view(-1, 1, 1)
        s = torch.as_tensor(stds, dtype=image.dtype).view(-1, 1, 1)
        return means, stds, (image - m) / s

    def read_dem(self, path):
        """dataset/Dataset.py:1131-1170: the no-data value is a large float, not NaN"""
        cfg = self.configs
        a, meta = geotiff.read(path, dtype=np.float32, squeeze=False)
        nodata = meta["nodata"]
        if nodata is not None and nodata == nodata:
            a = np.where(a == np.float32(nodata), np.float32("nan"), a)
        a = np.stack([fill_nodata_nearest(b) for b in a])
        if not cfg["dem"]:
            return torch.from_numpy(a)               # (read and dropped, as the reference does for a cell that ships a DEM)
        if cfg.get("slope"):
            out, mean, std = torch.from_numpy(slope_riserun(a[0], nodata)[None]), cfg["slc_slope_mean"], cfg["slc_slope_std"]
        else:
            out, mean, std = torch.from_numpy(a), cfg["slc_dem_mean"], cfg["slc_dem_std"]
        if cfg["scale_input"] is not None:
            out = (out - torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)) / torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)
        return out

    @staticmethod
    def pad_to_tile(img, value):
        """albumentations.PadIfNeeded(224, 224, border_mode=BORDER_CONSTANT, value=...) (:1173-1207): centred, the odd pixel goes to
        the bottom / right; img [C,H,W] or [H,W]"""
        h, w = img.shape[-2:]
        ph, pw = max(TILE - h, 0), max(TILE - w, 0)
        if not (ph or pw):
            return img
        pads = [(0, 0)] * (img.ndim - 2) + [(ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2)]
        return np.pad(img, pads, mode="constant", constant_values=value)

    def sample_files(self, index):
        sample = self._sample(index)
        return sample, _tile_files(os.path.join(self.root_path, sample["path"]))

    def __getitem__(self, idx):
        sample, files = self.sample_files(idx)
        cfg = self.configs

        def date(prefix):
            a = geotiff.read(files[prefix], dtype=np.float32, squeeze=False)[0]
            if cfg.get("uint8"):
                a = a / a.max()
                a = (a * 255).astype(np.uint8)
            return a
        flood, sec1, sec2 = date("MS1"), date("SL1"), date("SL2")
        mask = geotiff.read(files["MK0_MLU"])[0] if "MK0_MLU" in files else None
        dem = self.read_dem(files["MK0_DEM"]) if "MK0_DEM" in files else None
        if flood.shape != (4, TILE, TILE) or sec1.shape != (4, TILE, TILE) or sec2.shape != (4, TILE, TILE):
            fill = int(flood.mean())
            flood, sec1, sec2 = (self.pad_to_tile(a, fill) for a in (flood, sec1, sec2))
            if mask is not None:
                mask = self.pad_to_tile(mask, 3)
        clz, activation = sample["clz"], sample["activation"]
        tail = ((dem,) if cfg["dem"] else ()) + (clz, activation)
        if cfg["scale_input"] == "normalize":
            flood, sec1, sec2 = (torch.from_numpy(a).float() for a in (flood, sec1, sec2))
            mask = torch.from_numpy(np.asarray(mask)).long()
            m0, s0, flood = self.normalize(flood)
            m1, s1, sec1 = self.normalize(sec1)
            m2, s2, sec2 = self.normalize(sec2)
            return (m0, s0, flood, mask, m1, s1, sec1, m2, s2, sec2) + tail
        return (flood, mask, sec1, sec2) + tail


class _Ragged(Exception):
    pass


def _lib_error():
    from ._lib import KsmiError
    return KsmiError


def _collate_to(items, dev, sharded):
    """torch's default collate of per-sample tuples, tensors moved to `dev` (the slow path of a batch the native reader refuses)"""
    from torch.utils.data import default_collate
    out = [t.to(dev) if torch.is_tensor(t) and t.dim() > 1 else t for t in default_collate(items)]
    return ShardedBatch(out) if sharded else tuple(out)


class ShardedBatch(tuple):
    """a collated batch that already holds only this rank's samples (distributed.shard_batch passes it through)"""


class TileBatchLoader:
    """Batch-level loader for `Dataset` (GRD) with scale_input == "normalize": the collated tuple of dataset/Dataset.py:824-860 with
    the three dates decoded by ksmi_tile_batch_read into a pinned buffer, copied once and normalised on `device`.

    raw=True leaves the images as raw backscatter (NaNs included) for a model that normalises inside its first convolution
    (SNUNet_ECAM.set_input_pipeline); the per-channel scale lists of the tuple are what such a model needs.
    rank / world: this rank reads only its contiguous slice of every global batch (the slice distributed.shard_batch would cut).
    prefetch: batches decoded ahead by a producer thread (the native decode releases the GIL; its copy and preprocess run on a copy
    stream of their own), 0 = decode in the consumer's thread."""

    def __init__(self, dataset, batch_size, shuffle=False, drop_last=False, device="cuda", threads=8, raw=False, rank=0, world=1, seed=None,
                 prefetch=2):
        cfg = dataset.configs
        self.slc = isinstance(dataset, SLCDataset)
        if not isinstance(dataset, (Dataset, SLCDataset)) or cfg["scale_input"] != "normalize" or cfg.get("uint8") or cfg.get("slope"):
            raise ValueError("TileBatchLoader: Dataset / SLCDataset with scale_input 'normalize' (no uint8, no slope)")
        if not self.slc and cfg["clamp_input"] is None:
            raise ValueError("TileBatchLoader: clamp_input is required (the GPU preprocess clamps)")
        if batch_size % world:
            raise ValueError(f"global batch {batch_size} is not divisible by world size {world}")
        self.ds, self.bs, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.device, self.threads, self.raw, self.rank, self.world = torch.device(device), threads, raw, rank, world
        self.gen = random.Random(seed)
        self.dataset = dataset                      # (the trainers read loader.dataset.activations)
        self.nch = len(cfg["channels"])
        per = batch_size // world
        pin = self.device.type == "cuda"
        self.prefetch = max(0, int(prefetch))
        self.stage = torch.empty((per * ((12 if self.slc else 6) + 1 + (1 if cfg["dem"] else 0)), TILE, TILE), dtype=torch.float32, pin_memory=pin)
        self.copy_stream = torch.cuda.Stream(self.device) if pin else None

    def __len__(self):
        n = len(self.ds)
        return n // self.bs if self.drop_last else -(-n // self.bs)

    def _index_lists(self):
        order = list(range(len(self.ds)))
        if self.shuffle:
            self.gen.shuffle(order)
        for b in range(len(self)):
            idx = order[b * self.bs:(b + 1) * self.bs]
            n = len(idx)
            yield idx[n * self.rank // self.world:n * (self.rank + 1) // self.world]

    def __iter__(self):
        if not self.prefetch:
            for idx in self._index_lists():
                yield self.load(idx)
            return
        import queue
        import threading
        q, stop = queue.Queue(maxsize=self.prefetch), threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                for idx in self._index_lists():
                    if not put(self.load(idx)):
                        return
                put(None)
            except BaseException as e:          # noqa: BLE001  (handed to the consumer, which re-raises it)
                put(e)
        th = threading.Thread(target=produce, name="ksmi-tile-loader", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                if self.copy_stream is not None:                # produced on the copy stream, consumed on the caller's stream
                    cur = torch.cuda.current_stream(self.device)
                    for t in item:
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(cur)
                yield item
        finally:
            stop.set()
            th.join()

    def load(self, indices):
        """one collated batch; device work (copy, preprocess) runs on the loader's copy stream and is complete on return"""
        if self.copy_stream is None:
            return self._load(indices)
        with torch.cuda.stream(self.copy_stream):
            out = self._load(indices)
            self.copy_stream.synchronize()                      # also frees the staging buffer for the next batch
        return out

    def _load(self, indices):
        if self.slc:
            return self._load_slc(indices)
        return self._load_grd(indices)

    def _load_slc(self, indices):
        """4-band tiles (dataset/Dataset.py:1087-1228): [n x (MS1, SL1, SL2) x 4 bands][DEM][masks] in the staging buffer, Normalize on the
        device.  A batch with a ragged tile (smaller than 224 x 224: padded by the reference, :1173-1207) takes the per-sample path."""
        from .data import preprocess_gpu
        ds, cfg, dev, n = self.ds, self.ds.configs, self.device, len(indices)
        dem = bool(cfg["dem"])
        samples, files = zip(*[ds.sample_files(i) for i in indices]) if n else ((), ())
        stage = self.stage
        try:
            sar = geotiff.read_batch([f[p] for f in files for p in ("MS1", "SL1", "SL2")], TILE, TILE, out=stage[:12 * n].view(3 * n, 4, TILE, TILE),
                                     threads=self.threads, bands=4).view(n, 12, TILE, TILE)
            k = 12 * n
            dems = None
            if dem:
                dtl = geotiff.read_batch([f["MK0_DEM"] for f in files], TILE, TILE, out=stage[k:k + n], threads=self.threads)
                nodata = geotiff.info(files[0]["MK0_DEM"])["nodata"] if n else None
                if nodata is not None and nodata == nodata:
                    dtl[dtl == float(np.float32(nodata))] = float("nan")
                dems = geotiff.fill_nodata(dtl, threads=self.threads).view(n, 1, TILE, TILE).to(dev, non_blocking=True)
                k += n
            with_mask = [j for j, f in enumerate(files) if "MK0_MLU" in f]
            if len(with_mask) != n:
                raise _Ragged()                      # (the SLC class has no mask default: leave such a cell to __getitem__)
            mask = geotiff.read_batch([f["MK0_MLU"] for f in files], TILE, TILE, out=stage[k:k + n], threads=self.threads).to(dev, non_blocking=True).long()
        except (_Ragged, _lib_error()) as e:
            if not isinstance(e, _Ragged) and "expected" not in str(e):
                raise
            items = [ds[i] for i in indices]
            return _collate_to(items, dev, self.world > 1)
        sar = sar.to(dev, non_blocking=True)
        means, stds = cfg["slc_mean"], cfg["slc_std"]
        dates = [sar[:, a:a + 4].contiguous() for a in (0, 4, 8)]
        if not self.raw:
            if dev.type != "cuda":
                raise RuntimeError("TileBatchLoader normalises on the GPU (raw=True hands out raw tiles on any device)")
            dates = [preprocess_gpu(d, means, stds, -1.0) for d in dates]
        flood, sec1, sec2 = dates
        sv = lambda v: [torch.full((n,), float(x), dtype=torch.float64) for x in v]
        out = [sv(means), sv(stds), flood, mask, sv(means), sv(stds), sec1, sv(means), sv(stds), sec2]
        if dem:
            out.append((dems - torch.as_tensor(cfg["slc_dem_mean"], dtype=torch.float32, device=dev).view(1, -1, 1, 1)) /
                       torch.as_tensor(cfg["slc_dem_std"], dtype=torch.float32, device=dev).view(1, -1, 1, 1))
        out += [torch.tensor([s["clz"] for s in samples], dtype=torch.int64), torch.tensor([s["activation"] for s in samples], dtype=torch.int64)]
        return ShardedBatch(out) if self.world > 1 else tuple(out)

    def _load_grd(self, indices):
        from .data import preprocess_gpu
        ds, cfg, dev, n = self.ds, self.ds.configs, self.device, len(indices)
        dem = bool(cfg["dem"])
        samples, files = zip(*[ds.sample_files(i) for i in indices]) if n else ((), ())
        # staging order: [n x 6 SAR tiles][DEM tiles][label masks] -> the SAR block is the source of the one host-to-device copy
        paths = [f[p] for f in files for p in SAR_PREFIXES]
        if dem:
            paths += [f["MK0_DEM"] for f in files]
        with_mask = [j for j, f in enumerate(files) if "MK0_MLU" in f]           # a cell without MK0_MLU: all zeros (Dataset.py:787-789)
        paths += [files[j]["MK0_MLU"] for j in with_mask]
        stage = self.stage[:len(paths)]
        geotiff.read_batch(paths, TILE, TILE, out=stage, threads=self.threads)
        sar = stage[:6 * n].view(n, 6, TILE, TILE).to(dev, non_blocking=True)
        k = 6 * n
        dems = None
        if dem:                                                   # gaps filled on the host (in the staging buffer), standardised on the device
            dems = geotiff.fill_nodata(stage[k:k + n], threads=self.threads).view(n, 1, TILE, TILE).to(dev, non_blocking=True)
            k += n
        mask = stage[k:k + len(with_mask)].to(dev, non_blocking=True).long()
        if len(with_mask) != n:
            full = torch.zeros((n, TILE, TILE), dtype=torch.int64, device=dev)
            full[torch.tensor(with_mask, dtype=torch.int64, device=dev)] = mask
            mask = full
        ch = cfg["channels"]
        if set(ch) == {"vv", "vh", "vh/vv"}:
            dates = [torch.stack((sar[:, a], sar[:, a + 1], sar[:, a + 1] / (sar[:, a] + 1e-7)), 1) for a in (0, 2, 4)]
        elif set(ch) == {"vv", "vh"}:
            dates = [sar[:, a:a + 2] for a in (0, 2, 4)]
        elif ch == ["vh"]:
            dates = [sar[:, a + 1:a + 2] for a in (0, 2, 4)]
        else:
            raise ValueError(f"unsupported channels {ch}")
        means, stds = cfg["data_mean"], cfg["data_std"]
        if not self.raw:
            if dev.type != "cuda":
                raise RuntimeError("TileBatchLoader normalises on the GPU (raw=True hands out raw tiles on any device)")
            dates = [preprocess_gpu(d, means, stds, cfg["clamp_input"]) for d in dates]
        else:
            dates = [d.contiguous() for d in dates]
        flood, pre1, pre2 = dates
        sv = lambda v: [torch.full((n,), float(x), dtype=torch.float64) for x in v]
        out = [sv(means), sv(stds), flood, mask, sv(means), sv(stds), pre1, sv(means), sv(stds), pre2]
        if dem:
            out.append((dems - torch.as_tensor(cfg["dem_mean"], dtype=torch.float32, device=dev).view(1, -1, 1, 1)) /
                       torch.as_tensor(cfg["dem_std"], dtype=torch.float32, device=dev).view(1, -1, 1, 1))
        out += [torch.tensor([s["clz"] for s in samples], dtype=torch.int64), torch.tensor([s["activation"] for s in samples], dtype=torch.int64)]
        return ShardedBatch(out) if self.world > 1 else tuple(out)
