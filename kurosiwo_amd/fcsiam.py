"""FC-Siam-conc / FC-Siam-diff (row N2 of SURVEY.md §8: the other siamese change-detection baselines of the reference) on the
hand-written gfx950 kernels of the SNUNet / UNet paths.

Drop-in for /root/reference/models/siam_conc.py:13-177 (`SiamUnet_conc`) and siam_diff.py:13-172 (`SiamUnet_diff`): same constructor
(`input_nbr`, `label_nbr`), the same state-dict keys in the same order, `model(x1, x2) -> [B, label_nbr, H, W]` softmax map
(conc: `nn.Softmax(dim=1)`) or log-softmax map (diff: `nn.LogSoftmax(dim=1)`, siam_diff.py:93), train / eval BatchNorm semantics (the shared encoder normalises each date with its own batch
statistics and updates the running statistics twice per step, date 1 first), `nn.Dropout2d(p=0.2)` after every BN + ReLU.

MI355X-first: NHWC bf16 activations; every 3x3 convolution, the stride-1 transposed convolutions of the decoder (= the input
gradient form of a convolution: flipped taps, reduction-major weights), the stride-2 `upconv` (four 2x2 phase convolutions) and all
weight / input gradients run on the implicit-GEMM MFMA kernels with the concat of the skips fused into the operand walk;
BN-apply + ReLU + Dropout2d is one elementwise kernel whose (sample, channel) mask comes from the counter-based stream
(csrc/common.h) and is read back from `out > 0` in the backward pass.
"""
import math
from collections import OrderedDict

import torch

from . import _lib
from .arena import ArenaModule, PlanFn
from .runtime import require_gpu

# encoder: (name, Cout) per stage; decoder: per stage the upconv channel count and the (name, Cout) chain (siam_conc.py:19-93)
ENCODER = ((("11", 16), ("12", 16)), (("21", 32), ("22", 32)), (("31", 64), ("32", 64), ("33", 64)), (("41", 128), ("42", 128), ("43", 128)))
DECODER = ((4, 128, (("43d", 128), ("42d", 128), ("41d", 64))), (3, 64, (("33d", 64), ("32d", 64), ("31d", 32))),
           (2, 32, (("22d", 32), ("21d", 16))), (1, 16, (("12d", 16),)))
DROP2D = 0.2


def fcsiam_specs(input_nbr, label_nbr, diff):
    p, b, c = OrderedDict(), OrderedDict(), OrderedDict()

    def bn(name, ch):
        p[f"{name}.weight"] = (ch,)
        p[f"{name}.bias"] = (ch,)
        b[f"{name}.running_mean"] = (ch,)
        b[f"{name}.running_var"] = (ch,)
        c[f"{name}.num_batches_tracked"] = ()
    cin = input_nbr
    for stage in ENCODER:
        for name, co in stage:
            p[f"conv{name}.weight"] = (co, cin, 3, 3)
            p[f"conv{name}.bias"] = (co,)
            bn(f"bn{name}", co)
            cin = co
    for lvl, cu, chain in DECODER:
        p[f"upconv{lvl}.weight"] = (cu, cu, 3, 3)                 # ConvTranspose2d: [in][out][3][3]
        p[f"upconv{lvl}.bias"] = (cu,)
        ci = cu * (2 if diff else 3)                              # cat(up, |s1 - s2|) or cat(up, s1, s2)
        for name, co in chain:
            p[f"conv{name}.weight"] = (ci, co, 3, 3)
            p[f"conv{name}.bias"] = (co,)
            bn(f"bn{name}", co)
            ci = co
    p["conv11d.weight"] = (16, label_nbr, 3, 3)
    p["conv11d.bias"] = (label_nbr,)
    return p, b, c


class _SiamUnet(ArenaModule):
    diff = False

    def __init__(self, input_nbr, label_nbr, precision="bf16"):
        super().__init__()
        if label_nbr > 8:
            raise _lib.KsmiError("SiamUnet (HIP): label_nbr <= 8")
        self.input_nbr, self.label_nbr, self.precision = input_nbr, label_nbr, precision
        self.drop2d = DROP2D                                      # nn.Dropout2d(p=0.2) everywhere; 0 switches the layers off
        ps, bs, cs = fcsiam_specs(input_nbr, label_nbr, self.diff)
        self._setup_arena(ps, bs, cs)
        with torch.no_grad():                                     # PyTorch defaults: kaiming_uniform(a=sqrt 5) = U(+-1/sqrt(fan_in))
            for key, shp in self._pspec.items():
                v = self._p(key).view(shp)
                if key.startswith("bn"):
                    v.fill_(1.0 if key.endswith("weight") else 0.0)
                    continue
                w = shp if len(shp) == 4 else self._pspec[key[:-4] + "weight"]
                fan_in = w[1] * 9            # torch takes weight.size(1) * k*k for Conv2d and ConvTranspose2d alike
                v.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
            for key in self._bspec:
                self._b(key).fill_(1.0 if key.endswith("running_var") else 0.0)

    def plan(self, B, H, W, training, with_backward):
        self._ensure_arena()
        key = (B, H, W, self.act_dtype(), bool(training), bool(with_backward), self.drop2d if training else None)
        if key not in self._plans:
            from .fcsiam_plan import FCSiamPlan
            self._plans[key] = FCSiamPlan(self, B, H, W, self.act_dtype(), training, with_backward)
        return self._plans[key]

    def forward(self, x1, x2):
        require_gpu(x1)
        if x1.shape != x2.shape or x1.dim() != 4 or x1.shape[1] != self.input_nbr or x1.shape[2] % 16 or x1.shape[3] % 16:
            raise ValueError(f"expected two [B,{self.input_nbr},H,W] tensors with H, W multiples of 16, got {tuple(x1.shape)} {tuple(x2.shape)}")
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        plan = self.plan(x1.shape[0], x1.shape[2], x1.shape[3], self.training, want_grad)
        x1, x2 = x1.contiguous().float(), x2.contiguous().float()
        if not want_grad:
            return plan.run_forward(x1, x2).clone()
        if self._anchor is None or self._anchor.device != x1.device:
            self._anchor = torch.zeros(1, device=x1.device, requires_grad=True)
        return PlanFn.apply(self._anchor, self, plan, x1, x2)


class SiamUnet_conc(_SiamUnet):
    """siam_conc.py:13-177"""
    diff = False


class SiamUnet_diff(_SiamUnet):
    """siam_diff.py:13-172"""
    diff = True
