"""Oracle (TEST INFRASTRUCTURE) for row N2 of SURVEY.md §8: FC-Siam-conc / FC-Siam-diff.

A functional, state-dict driven fp32 restatement on stock PyTorch-CPU ops of /root/reference/models/siam_conc.py:97-177 and
siam_diff.py:95-172 (Daudt et al., ICIP 2018): shared encoder run on each date with its OWN BatchNorm batch statistics (two
running-statistics updates per step, date 1 first), conv -> BN -> ReLU -> Dropout2d(0.2) everywhere, 2x2 max pooling, stride-2
ConvTranspose2d upsampling, skip concat (conc: both dates; diff: |date1 - date2|), ConvTranspose2d(k3, p1) decoder convolutions,
nn.Softmax(dim=1) output (conc) / nn.LogSoftmax(dim=1) (diff).  The ReplicationPad2d of the decoder is the identity for sizes divisible by 16.

Dropout2d draws come from the counter-based stream of oracle/rng_ref.py (`stream=`): mask index b*C + c of site 2*layer + date
(decoder layers: date 0), exactly what kurosiwo_amd/fcsiam_plan.py / csrc/stochastic.hip regenerate.  Pinned to the real reference
by tests/golden/fcsiam_{conc,diff}.npz (oracle/gen_golden.py:gen_fcsiam runs the reference's own modules on the same draws).
Only tests/, smoke() and bench.py's cpu_baseline may import this module.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

ENCODER = ((("11", 16), ("12", 16)), (("21", 32), ("22", 32)), (("31", 64), ("32", 64), ("33", 64)), (("41", 128), ("42", 128), ("43", 128)))
DECODER = ((4, 128, (("43d", 128), ("42d", 128), ("41d", 64))), (3, 64, (("33d", 64), ("32d", 64), ("31d", 32))),
           (2, 32, (("22d", 32), ("21d", 16))), (1, 16, (("12d", 16),)))
BN_EPS, BN_MOMENTUM, DROP2D = 1e-5, 0.1, 0.2
# BN layers in definition order: the site of a layer's Dropout2d is 2 * index + date
LAYERS = [n for st in ENCODER for n, _ in st] + [n for _, _, ch in DECODER for n, _ in ch]


def state_dict_spec(input_nbr=2, label_nbr=3, diff=False):
    s = OrderedDict()

    def bn(name, ch):
        s[f"{name}.weight"], s[f"{name}.bias"] = (ch,), (ch,)
        s[f"{name}.running_mean"], s[f"{name}.running_var"], s[f"{name}.num_batches_tracked"] = (ch,), (ch,), ()
    cin = input_nbr
    for stage in ENCODER:
        for name, co in stage:
            s[f"conv{name}.weight"], s[f"conv{name}.bias"] = (co, cin, 3, 3), (co,)
            bn(f"bn{name}", co)
            cin = co
    for lvl, cu, chain in DECODER:
        s[f"upconv{lvl}.weight"], s[f"upconv{lvl}.bias"] = (cu, cu, 3, 3), (cu,)
        ci = cu * (2 if diff else 3)
        for name, co in chain:
            s[f"conv{name}.weight"], s[f"conv{name}.bias"] = (ci, co, 3, 3), (co,)
            bn(f"bn{name}", co)
            ci = co
    s["conv11d.weight"], s["conv11d.bias"] = (16, label_nbr, 3, 3), (label_nbr,)
    return s


def new_state_dict(input_nbr=2, label_nbr=3, diff=False):
    sd = OrderedDict()
    for k, shp in state_dict_spec(input_nbr, label_nbr, diff).items():
        sd[k] = torch.zeros(shp, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
    return sd


def is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def _bn(sd, key, x, training, stats):
    """nn.BatchNorm2d; `stats` carries the running statistics across the two dates of a step"""
    if not training:
        return F.batch_norm(x, sd[f"{key}.running_mean"], sd[f"{key}.running_var"], sd[f"{key}.weight"], sd[f"{key}.bias"], False, 0.0, BN_EPS)
    rm = stats.get(f"{key}.running_mean", sd[f"{key}.running_mean"]).detach().clone()
    rv = stats.get(f"{key}.running_var", sd[f"{key}.running_var"]).detach().clone()
    y = F.batch_norm(x, rm, rv, sd[f"{key}.weight"], sd[f"{key}.bias"], True, BN_MOMENTUM, BN_EPS)
    stats[f"{key}.running_mean"], stats[f"{key}.running_var"] = rm, rv
    stats[f"{key}.num_batches_tracked"] = stats.get(f"{key}.num_batches_tracked", sd[f"{key}.num_batches_tracked"]) + 1
    return y


def _unit(sd, name, x, training, stats, stream, date, transposed, inter=None, tag=None):
    w, b = sd[f"conv{name}.weight"], sd[f"conv{name}.bias"]
    z = F.conv_transpose2d(x, w, b, padding=1) if transposed else F.conv2d(x, w, b, padding=1)
    y = F.relu(_bn(sd, f"bn{name}", z, training, stats))
    if training and stream is not None:
        from . import rng_ref as G
        B, Cc = y.shape[:2]
        m = G.scale_mask(stream[0], stream[1], 2 * LAYERS.index(name) + date, stream[2], 0, (B, Cc))
        y = y * torch.from_numpy(m)[:, :, None, None]
    if inter is not None:
        inter[tag or name] = y
    return y


def forward(sd, x1, x2, diff=False, training=False, stats=None, stream=None, inter=None):
    """stream = (seed, step, p) or None; returns the model output (conc: softmax map, diff: log-softmax map)"""
    stats = {} if stats is None else stats
    skips = []
    for date, x in enumerate((x1, x2)):
        sk = []
        for stage in ENCODER:
            for name, _ in stage:
                x = _unit(sd, name, x, training, stats, stream, date, False, inter, f"{name}_{date + 1}")
            sk.append(x)
            x = F.max_pool2d(x, 2, 2)
        skips.append(sk)
    y = x                                                         # x4p of date 2 (siam_conc.py:148-150)
    for lvl, cu, chain in DECODER:
        y = F.conv_transpose2d(y, sd[f"upconv{lvl}.weight"], sd[f"upconv{lvl}.bias"], stride=2, padding=1, output_padding=1)
        s1, s2 = skips[0][lvl - 1], skips[1][lvl - 1]
        y = torch.cat((y, torch.abs(s1 - s2)), 1) if diff else torch.cat((y, s1, s2), 1)
        for name, _ in chain:
            y = _unit(sd, name, y, training, stats, stream, 0, True, inter)
    logits = F.conv_transpose2d(y, sd["conv11d.weight"], sd["conv11d.bias"], padding=1)
    if inter is not None:
        inter["logits"] = logits
    return torch.log_softmax(logits, dim=1) if diff else torch.softmax(logits, dim=1)   # siam_diff.py:93 / siam_conc.py:93


def loss_and_grads(sd, x1, x2, labels, diff=False, weights=(1.0, 1.0, 1.0), stream=None):
    """One train-mode forward/backward with the reference's CD criterion applied to the model output (the softmax map)."""
    from .snunet_ref import torch_ce_dice
    params = {k: (v.detach().clone().requires_grad_(True) if not is_buffer(k) else v) for k, v in sd.items()}
    stats = {}
    out = forward(params, x1, x2, diff, True, stats, stream)
    loss = torch_ce_dice(out, labels, weights, True)
    total = loss[0] if isinstance(loss, (tuple, list)) else loss
    total.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items() if not is_buffer(k)}
    return out.detach(), float(total.detach()), grads, stats
