"""Oracle (TEST INFRASTRUCTURE) for row U1 of SURVEY.md §8(a): smp.Unet(resnet18) -- PARITY UNPINNED.

The reference builds this model from `segmentation_models_pytorch==0.3.2` (models/model_utilities.py:110-123, requirements.txt:14),
a third-party dependency whose source is NOT under /root/reference and is not installed here; its ImageNet encoder weights would
need the network.  This file restates the PUBLISHED architecture of smp 0.3.2 `Unet(encoder_name="resnet18", encoder_weights=None,
in_channels=c, classes=n)`:

  encoder  torchvision ResNet-18 without avgpool/fc: conv1 7x7 s2 p3 (no bias) -> bn1 -> relu | maxpool 3x3 s2 p1 -> layer1..4
           (BasicBlock: conv3x3(s) -> bn -> relu -> conv3x3 -> bn -> (+ identity | 1x1(s) conv + bn) -> relu), features at strides
           1 (input), 2, 4, 8, 16, 32 with channels (c, 64, 64, 128, 256, 512)
  decoder  UnetDecoder(decoder_channels=(256, 128, 64, 32, 16), use_batchnorm=True, center=False): 5 x DecoderBlock =
           nearest x2 -> cat(skip) -> Conv3x3(no bias)-BN-ReLU -> Conv3x3(no bias)-BN-ReLU
  head     Conv2d(16, classes, 3, padding=1)

No golden vector exists for it (nothing to import); the state-dict key names follow smp / torchvision as published.  The GPU tests
compare the HIP path with this restatement only.  Only tests/ import this module.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
LAYERS = ((64, 1), (128, 2), (256, 2), (512, 2))
DECODER_CHANNELS = (256, 128, 64, 32, 16)


def _bn_spec(s, name, c):
    s[f"{name}.weight"] = (c,)
    s[f"{name}.bias"] = (c,)
    s[f"{name}.running_mean"] = (c,)
    s[f"{name}.running_var"] = (c,)
    s[f"{name}.num_batches_tracked"] = ()


def unet_state_dict_spec(in_channels=2, classes=3):
    s = OrderedDict()
    s["encoder.conv1.weight"] = (64, in_channels, 7, 7)
    _bn_spec(s, "encoder.bn1", 64)
    cin = 64
    for li, (c, stride) in enumerate(LAYERS):
        for bi in range(2):
            p = f"encoder.layer{li + 1}.{bi}"
            s[f"{p}.conv1.weight"] = (c, cin, 3, 3)
            _bn_spec(s, f"{p}.bn1", c)
            s[f"{p}.conv2.weight"] = (c, c, 3, 3)
            _bn_spec(s, f"{p}.bn2", c)
            if bi == 0 and (stride != 1 or cin != c):
                s[f"{p}.downsample.0.weight"] = (c, cin, 1, 1)
                _bn_spec(s, f"{p}.downsample.1", c)
            cin = c
    enc = (512, 256, 128, 64, 64)
    ins = (enc[0],) + DECODER_CHANNELS[:-1]
    skips = enc[1:] + (0,)
    for i, (ci, cs, co) in enumerate(zip(ins, skips, DECODER_CHANNELS)):
        p = f"decoder.blocks.{i}"
        s[f"{p}.conv1.0.weight"] = (co, ci + cs, 3, 3)
        _bn_spec(s, f"{p}.conv1.1", co)
        s[f"{p}.conv2.0.weight"] = (co, co, 3, 3)
        _bn_spec(s, f"{p}.conv2.1", co)
    s["segmentation_head.0.weight"] = (classes, DECODER_CHANNELS[-1], 3, 3)
    s["segmentation_head.0.bias"] = (classes,)
    return s


def new_state_dict(in_channels=2, classes=3):
    sd = OrderedDict()
    for k, shp in unet_state_dict_spec(in_channels, classes).items():
        sd[k] = torch.zeros(shp, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
    return sd


def is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def _bn(sd, key, x, training, new_stats):
    if not training:
        return F.batch_norm(x, sd[f"{key}.running_mean"], sd[f"{key}.running_var"], sd[f"{key}.weight"], sd[f"{key}.bias"], False, BN_MOMENTUM, BN_EPS)
    rm, rv = sd[f"{key}.running_mean"].detach().clone(), sd[f"{key}.running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[f"{key}.weight"], sd[f"{key}.bias"], True, BN_MOMENTUM, BN_EPS)
    if new_stats is not None:
        new_stats[f"{key}.running_mean"], new_stats[f"{key}.running_var"] = rm, rv
    return y


def _relu(x, masks, name):
    if masks is None or name not in masks:
        return F.relu(x)
    return x * masks[name]


def unet_forward(sd, x, training=False, new_stats=None, inter=None, masks=None):
    """`masks`: optional {name: 0/1 tensor} pinning the ReLU active sets (see oracle/vit_ref.py)."""
    feats = [x]
    t = F.conv2d(x, sd["encoder.conv1.weight"], None, stride=2, padding=3)
    t = _relu(_bn(sd, "encoder.bn1", t, training, new_stats), masks, "stem")
    feats.append(t)
    t = F.max_pool2d(t, 3, 2, 1)
    cin = 64
    for li, (c, stride) in enumerate(LAYERS):
        for bi in range(2):
            p = f"encoder.layer{li + 1}.{bi}"
            s_ = stride if bi == 0 else 1
            idn = t
            o = F.conv2d(t, sd[f"{p}.conv1.weight"], None, stride=s_, padding=1)
            o = _relu(_bn(sd, f"{p}.bn1", o, training, new_stats), masks, f"{p}.r1")
            o = _bn(sd, f"{p}.bn2", F.conv2d(o, sd[f"{p}.conv2.weight"], None, padding=1), training, new_stats)
            if f"{p}.downsample.0.weight" in sd:
                idn = _bn(sd, f"{p}.downsample.1", F.conv2d(t, sd[f"{p}.downsample.0.weight"], None, stride=s_), training, new_stats)
            t = _relu(o + idn, masks, f"{p}.out")
            cin = c
        feats.append(t)
    if inter is not None:
        for i, f in enumerate(feats):
            inter[f"f{i}"] = f
    skips = feats[1:][::-1]                 # f5, f4, f3, f2, f1
    y = skips[0]
    for i in range(5):
        p = f"decoder.blocks.{i}"
        y = F.interpolate(y, scale_factor=2, mode="nearest")
        if i + 1 < len(skips):
            y = torch.cat([y, skips[i + 1]], dim=1)
        y = _relu(_bn(sd, f"{p}.conv1.1", F.conv2d(y, sd[f"{p}.conv1.0.weight"], None, padding=1), training, new_stats), masks, f"{p}.r1")
        y = _relu(_bn(sd, f"{p}.conv2.1", F.conv2d(y, sd[f"{p}.conv2.0.weight"], None, padding=1), training, new_stats), masks, f"{p}.r2")
        if inter is not None:
            inter[f"d{i}"] = y
    return F.conv2d(y, sd["segmentation_head.0.weight"], sd["segmentation_head.0.bias"], padding=1)


def loss_and_grads(sd, x, labels, weights=None, masks=None):
    params = {k: (v.detach().clone().requires_grad_(True) if not is_buffer(k) else v) for k, v in sd.items()}
    new_stats = {}
    logits = unet_forward(params, x, training=True, new_stats=new_stats, masks=masks)
    w = None if weights is None else torch.tensor(list(weights), dtype=logits.dtype)
    loss = F.cross_entropy(logits, labels, weight=w, ignore_index=3)
    loss.backward()
    return logits.detach(), float(loss.detach()), {k: p.grad for k, p in params.items() if not is_buffer(k)}, new_stats
