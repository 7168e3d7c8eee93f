"""Oracle (TEST INFRASTRUCTURE) for row N2 of SURVEY.md §8: BIT-CD in the shipped configuration `net_G = base_resnet18`.

Functional, state-dict driven fp32 restatement on stock PyTorch-CPU ops of /root/reference/models/bit_cd.py: `ResNet.forward` :763-778,
`forward_single` :780-797 (conv1 7x7 s2 -> bn1 -> relu -> maxpool 3x3 s2 -> layer1..4 with strides (1, 2, 1, 1): layer3 / layer4 are
"dilated" but `BasicBlock` resets the dilation, :97-98 -> nearest x2 -> conv_pred 3x3), |f1 - f2|, bilinear x4 (nn.Upsample default
align_corners=False), `TwoLayerConv2d` classifier :416-424.  The shared backbone sees each date on its own (two BatchNorm updates per
step, date 1 first).  Pinned to the real reference by tests/golden/bitcd.npz.  Only tests/, smoke() and bench.py's cpu_baseline may import it.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

LAYERS = ((64, 1), (128, 2), (256, 1), (512, 1))
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def state_dict_spec(input_nc=2, output_nc=3):
    s = OrderedDict()

    def bn(name, ch):
        s[f"{name}.weight"], s[f"{name}.bias"] = (ch,), (ch,)
        s[f"{name}.running_mean"], s[f"{name}.running_var"], s[f"{name}.num_batches_tracked"] = (ch,), (ch,), ()
    s["resnet.conv1.weight"] = (64, input_nc, 7, 7)
    bn("resnet.bn1", 64)
    cin = 64
    for li, (ch, stride) in enumerate(LAYERS):
        for bi in range(2):
            k = f"resnet.layer{li + 1}.{bi}"
            s[f"{k}.conv1.weight"] = (ch, cin, 3, 3)
            bn(f"{k}.bn1", ch)
            s[f"{k}.conv2.weight"] = (ch, ch, 3, 3)
            bn(f"{k}.bn2", ch)
            if bi == 0 and (stride != 1 or cin != ch):
                s[f"{k}.downsample.0.weight"] = (ch, cin, 1, 1)
                bn(f"{k}.downsample.1", ch)
            cin = ch
    s["resnet.fc.weight"], s["resnet.fc.bias"] = (1000, 512), (1000,)
    s["classifier.0.weight"] = (32, 32, 3, 3)
    bn("classifier.1", 32)
    s["classifier.3.weight"], s["classifier.3.bias"] = (output_nc, 32, 3, 3), (output_nc,)
    s["conv_pred.weight"], s["conv_pred.bias"] = (32, 512, 3, 3), (32,)
    return s


def new_state_dict(input_nc=2, output_nc=3):
    sd = OrderedDict()
    for k, shp in state_dict_spec(input_nc, output_nc).items():
        sd[k] = torch.zeros(shp, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
    return sd


def is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def _bn(sd, key, x, training, stats):
    if not training:
        return F.batch_norm(x, sd[f"{key}.running_mean"], sd[f"{key}.running_var"], sd[f"{key}.weight"], sd[f"{key}.bias"], False, 0.0, BN_EPS)
    rm = stats.get(f"{key}.running_mean", sd[f"{key}.running_mean"]).detach().clone()
    rv = stats.get(f"{key}.running_var", sd[f"{key}.running_var"]).detach().clone()
    y = F.batch_norm(x, rm, rv, sd[f"{key}.weight"], sd[f"{key}.bias"], True, BN_MOMENTUM, BN_EPS)
    stats[f"{key}.running_mean"], stats[f"{key}.running_var"] = rm, rv
    stats[f"{key}.num_batches_tracked"] = stats.get(f"{key}.num_batches_tracked", sd[f"{key}.num_batches_tracked"]) + 1
    return y


def _block(sd, k, x, stride, training, stats):
    out = F.relu(_bn(sd, f"{k}.bn1", F.conv2d(x, sd[f"{k}.conv1.weight"], None, stride, 1), training, stats))
    out = _bn(sd, f"{k}.bn2", F.conv2d(out, sd[f"{k}.conv2.weight"], None, 1, 1), training, stats)
    if f"{k}.downsample.0.weight" in sd:
        x = _bn(sd, f"{k}.downsample.1", F.conv2d(x, sd[f"{k}.downsample.0.weight"], None, stride, 0), training, stats)
    return F.relu(out + x)


def forward_single(sd, x, training, stats, inter=None, tag=""):
    x = F.relu(_bn(sd, "resnet.bn1", F.conv2d(x, sd["resnet.conv1.weight"], None, 2, 3), training, stats))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (_, stride) in enumerate(LAYERS):
        for bi in range(2):
            x = _block(sd, f"resnet.layer{li + 1}.{bi}", x, stride if bi == 0 else 1, training, stats)
        if inter is not None:
            inter[f"layer{li + 1}{tag}"] = x
    x = F.interpolate(x, scale_factor=2)                                   # nn.Upsample(scale_factor=2): nearest
    x = F.conv2d(x, sd["conv_pred.weight"], sd["conv_pred.bias"], 1, 1)
    if inter is not None:
        inter[f"pred{tag}"] = x
    return x


def forward(sd, x1, x2, training=False, stats=None, inter=None):
    stats = {} if stats is None else stats
    f1 = forward_single(sd, x1, training, stats, inter, "_1")
    f2 = forward_single(sd, x2, training, stats, inter, "_2")
    x = torch.abs(f1 - f2)
    x = F.interpolate(x, scale_factor=4, mode="bilinear", align_corners=False)
    x = F.relu(_bn(sd, "classifier.1", F.conv2d(x, sd["classifier.0.weight"], None, 1, 1), training, stats))
    if inter is not None:
        inter["cls"] = x
    return F.conv2d(x, sd["classifier.3.weight"], sd["classifier.3.bias"], 1, 1)


def loss_and_grads(sd, x1, x2, labels, weights=(1.0, 1.0, 1.0)):
    from .snunet_ref import torch_ce_dice
    params = {k: (v.detach().clone().requires_grad_(True) if not is_buffer(k) else v) for k, v in sd.items()}
    stats = {}
    out = forward(params, x1, x2, True, stats)
    total = torch_ce_dice(out, labels, weights, True)
    total.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items() if not is_buffer(k)}
    return out.detach(), float(total.detach()), grads, stats
