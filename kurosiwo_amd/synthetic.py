"""Synthetic Sentinel-1 tile generator (replaces the reference's Dataset.__getitem__, which
needs the real Kuro Siwo archive + cv2/rioxarray).  Reproduces the *collated batch tuple* of
/root/reference/dataset/Dataset.py:824-860 exactly as the trainers unpack it
(training/change_detection_trainer.py:95-106), with value statistics following
SURVEY.md §8(d): Gamma speckle, clamp [0, 0.15] (Dataset.py:164-166), normalise with
data_mean/std (configs/train/data_config.json:16-17), water ellipses, 2 % invalid pixels.
"""
import torch

DATA_MEAN = (0.0953, 0.0264)
DATA_STD = (0.0427, 0.0215)
# SLC tiles carry 4 bands per date (dataset/Dataset.py:986-1228; configs/train/data_config.json slc_mean / slc_std)
SLC_MEAN = (0.022367, 39.242, 81.13, 0.043526)
SLC_STD = (1.2843, 25.6152, 58.0151, 1.2844)
DEM_MEAN, DEM_STD = 93.4313, 1410.8382


def _ellipse_mask(B, H, W, n, gen):
    yy = torch.arange(H).view(1, H, 1).float()
    xx = torch.arange(W).view(1, 1, W).float()
    m = torch.zeros((B, H, W), dtype=torch.bool)
    for _ in range(n):
        cy = torch.rand((B, 1, 1), generator=gen) * H
        cx = torch.rand((B, 1, 1), generator=gen) * W
        ry = 4 + torch.rand((B, 1, 1), generator=gen) * H / 4
        rx = 4 + torch.rand((B, 1, 1), generator=gen) * W / 4
        m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
    return m


def make_batch(B, H=224, W=224, seed=999, dem=False, channels=2, device="cpu"):
    """Returns the 12-tuple (13 with dem) of the reference's collated batch."""
    g = torch.Generator().manual_seed(seed)
    if channels > 2:                      # SLC: band statistics of the 4-band product; the speckle model is reused per band
        base_mean, base_std = (0.0953, 0.0264, 0.0953, 0.0264)[:channels], (0.0427, 0.0215, 0.0427, 0.0215)[:channels]
    else:
        base_mean, base_std = DATA_MEAN[:channels], DATA_STD[:channels]
    mean = torch.tensor(base_mean).view(1, channels, 1, 1)
    std = torch.tensor(base_std).view(1, channels, 1, 1)
    perm = _ellipse_mask(B, H, W, 2, g)
    flood = _ellipse_mask(B, H, W, 2, g) & ~perm
    mask = torch.zeros((B, H, W), dtype=torch.int64)
    mask[perm] = 1
    mask[flood] = 2
    mask[torch.rand((B, H, W), generator=g) < 0.02] = 3

    def date(water):
        k = 4.0
        gam = torch.distributions.Gamma(torch.tensor(k), torch.tensor(k))
        torch.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g)))
        sigma0 = gam.sample((B, channels, H, W)) * mean
        sigma0 = torch.where(water.unsqueeze(1), sigma0 * 0.2, sigma0)
        sigma0 = torch.nan_to_num(sigma0.clamp(0.0, 0.15))
        return ((sigma0 - mean) / std).float()
    post = date(perm | flood)
    pre1 = date(perm)
    pre2 = date(perm)
    sv = lambda v: [torch.full((B,), float(x), dtype=torch.float64) for x in (v if channels <= 2 else (SLC_MEAN if v is DATA_MEAN else SLC_STD))[:channels]]
    clz = torch.randint(1, 4, (B,), generator=g, dtype=torch.int64)
    activ = torch.randint(100, 600, (B,), generator=g, dtype=torch.int64)
    items = [sv(DATA_MEAN), sv(DATA_STD), post, mask, sv(DATA_MEAN), sv(DATA_STD), pre1, sv(DATA_MEAN), sv(DATA_STD), pre2]
    if dem:
        yy = torch.linspace(0, 3.14, H).view(1, 1, H, 1)
        xx = torch.linspace(0, 6.28, W).view(1, 1, 1, W)
        field = 200 + 150 * torch.sin(yy + torch.rand((B, 1, 1, 1), generator=g) * 6) * torch.cos(xx)
        items.append(((field - DEM_MEAN) / DEM_STD).float())
    items += [clz, activ]
    mv = lambda t: t.to(device) if torch.is_tensor(t) else t
    return tuple(mv(t) for t in items)


def cd_inputs(batch, inputs=("pre_event_1", "post_event"), dem=False):
    """The trainer's input assembly (training/change_detection_trainer.py:117-133)."""
    post, mask, pre1, pre2 = batch[2], batch[3], batch[6], batch[9]
    d = batch[10] if dem else None
    pick = {"pre_event_1": pre1, "pre_event_2": pre2, "post_event": post}
    xs = []
    for name in inputs:
        x = pick[name]
        xs.append(torch.cat((x, d), dim=1) if dem else x)
    return xs, mask


def seg_inputs(batch, inputs=("pre_event_1", "pre_event_2", "post_event"), dem=False):
    """The segmentation trainer's channel concat (training/segmentation_trainer.py:107-144):
    [post, (dem), pre1 and/or pre2] -> (image, mask)."""
    post, mask, pre1, pre2 = batch[2], batch[3], batch[6], batch[9]
    parts = [post]
    if dem:
        parts.append(batch[10])
    names = set(inputs)
    if names == {"post_event"}:
        pass
    elif names == {"pre_event_1", "post_event"}:
        parts.append(pre1)
    elif names == {"pre_event_2", "post_event"}:
        parts.append(pre2)
    elif names == {"pre_event_1", "pre_event_2", "post_event"}:
        parts += [pre1, pre2]
    else:
        raise ValueError('Invalid configuration for "inputs".')
    return torch.cat(parts, dim=1), mask
