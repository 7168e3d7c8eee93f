"""UNet with a ResNet-18 encoder (row U1 of SURVEY.md §8(a)) on hand-written gfx950 kernels.

The reference's segmentation baseline is `smp.Unet(encoder_name=backbone, encoder_weights=..., in_channels=num_channels,
classes=num_classes)` (/root/reference/models/model_utilities.py:110-123) from segmentation-models-pytorch 0.3.2, a third-party
package that is neither under /root/reference nor installed here: this class restates its published architecture (see
oracle/unet_ref.py for the layer list) with the same constructor keywords and the smp / torchvision state-dict key names
(182 keys, 14.33 M parameters at 2 input channels).  PARITY UNPINNED: there is nothing to import; ImageNet weights
(`encoder_weights="imagenet"`) would need the network and are refused.
"""
import math
from collections import OrderedDict

import torch

from . import _lib
from .arena import ArenaModule
from .runtime import require_gpu

LAYERS = ((64, 1), (128, 2), (256, 2), (512, 2))
DECODER_CHANNELS = (256, 128, 64, 32, 16)


def unet_specs(in_channels, classes):
    p, b, c = OrderedDict(), OrderedDict(), OrderedDict()

    def bn(name, ch):
        p[f"{name}.weight"] = (ch,)
        p[f"{name}.bias"] = (ch,)
        b[f"{name}.running_mean"] = (ch,)
        b[f"{name}.running_var"] = (ch,)
        c[f"{name}.num_batches_tracked"] = ()
    p["encoder.conv1.weight"] = (64, in_channels, 7, 7)
    bn("encoder.bn1", 64)
    cin = 64
    for li, (ch, stride) in enumerate(LAYERS):
        for bi in range(2):
            k = f"encoder.layer{li + 1}.{bi}"
            p[f"{k}.conv1.weight"] = (ch, cin, 3, 3)
            bn(f"{k}.bn1", ch)
            p[f"{k}.conv2.weight"] = (ch, ch, 3, 3)
            bn(f"{k}.bn2", ch)
            if bi == 0 and (stride != 1 or cin != ch):
                p[f"{k}.downsample.0.weight"] = (ch, cin, 1, 1)
                bn(f"{k}.downsample.1", ch)
            cin = ch
    enc = (512, 256, 128, 64, 64)
    ins = (enc[0],) + DECODER_CHANNELS[:-1]
    skips = enc[1:] + (0,)
    for i, (ci, cs, co) in enumerate(zip(ins, skips, DECODER_CHANNELS)):
        k = f"decoder.blocks.{i}"
        p[f"{k}.conv1.0.weight"] = (co, ci + cs, 3, 3)
        bn(f"{k}.conv1.1", co)
        p[f"{k}.conv2.0.weight"] = (co, co, 3, 3)
        bn(f"{k}.conv2.1", co)
    p["segmentation_head.0.weight"] = (classes, DECODER_CHANNELS[-1], 3, 3)
    p["segmentation_head.0.bias"] = (classes,)
    return p, b, c


class Unet(ArenaModule):
    def __init__(self, encoder_name="resnet18", encoder_depth=5, encoder_weights=None, decoder_use_batchnorm=True,
                 decoder_channels=DECODER_CHANNELS, in_channels=3, classes=1, activation=None, precision="bf16"):
        super().__init__()
        if encoder_name != "resnet18" or encoder_depth != 5 or tuple(decoder_channels) != DECODER_CHANNELS or not decoder_use_batchnorm:
            raise NotImplementedError("Unet (HIP): resnet18 encoder, depth 5, decoder (256,128,64,32,16) with BatchNorm (the reference's unet.json)")
        if encoder_weights is not None:
            raise _lib.KsmiError("Unet (HIP): pretrained encoder weights need the network; pass encoder_weights=None and load a state dict")
        if activation is not None or classes > 8:
            raise NotImplementedError("Unet (HIP): activation=None, classes <= 8")
        self.in_channels, self.classes, self.precision = in_channels, classes, precision
        ps, bs, cs = unet_specs(in_channels, classes)
        self._setup_arena(ps, bs, cs)
        with torch.no_grad():
            for key, shp in self._pspec.items():
                p = self._p(key).view(shp)
                if len(shp) == 1:
                    if key == "segmentation_head.0.bias" or key.endswith("bias"):
                        p.zero_()
                    else:
                        p.fill_(1.0)
                elif key.startswith("encoder."):
                    torch.nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")          # torchvision ResNet
                elif key.startswith("decoder."):
                    torch.nn.init.kaiming_uniform_(p, mode="fan_in", nonlinearity="relu")          # smp initialize_decoder
                else:
                    torch.nn.init.xavier_uniform_(p)                                               # smp initialize_head
            for key in self._bspec:
                self._b(key).fill_(1.0 if key.endswith("running_var") else 0.0)

    def plan(self, B, H, W, training, with_backward):
        self._ensure_arena()
        key = (B, H, W, self.act_dtype(), bool(training), bool(with_backward))
        if key not in self._plans:
            from .unet_plan import UnetPlan
            self._plans[key] = UnetPlan(self, B, H, W, self.act_dtype(), training, with_backward)
        return self._plans[key]

    def forward(self, x):
        require_gpu(x)
        if x.dim() != 4 or x.shape[1] != self.in_channels or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"expected [B,{self.in_channels},H,W] with H, W multiples of 32, got {tuple(x.shape)}")
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        plan = self.plan(x.shape[0], x.shape[2], x.shape[3], self.training, want_grad)
        x = x.contiguous().float()
        if not want_grad:
            return plan.run_forward(x).clone()
        if self._anchor is None or self._anchor.device != x.device:
            self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
        from .arena import PlanFn
        return PlanFn.apply(self._anchor, self, plan, x)
