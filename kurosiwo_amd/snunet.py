"""SNUNet-ECAM on hand-written gfx950 kernels (rows S1-S9 of SURVEY.md §8(a)).

Drop-in for the reference class (/root/reference/models/snunet.py:65-153): same
constructor, same state-dict keys (236), same ``model(xA, xB) -> logits [B,3,H,W]``
contract with autograd, ``.train()/.eval()``, BatchNorm running statistics.

MI355X-first design:
  * every parameter lives in one flat fp32 arena (and every gradient in a second one)
    so the optimiser step and the data-parallel all-reduce are single flat operations;
  * a *plan* (static launch list over preallocated NHWC activations) is built once per
    (batch, H, W, dtype, mode) and replayed -- capturable in a HIP graph;
  * torch.cat is never materialised (virtual concat inside the implicit-GEMM K loop),
    BatchNorm-apply + ReLU are fused into the consumer conv's operand load, BN statistics
    are produced by the conv epilogue.
There is no CPU/eager fallback: a CPU tensor raises.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from .runtime import require_gpu

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _blocks(n, c):
    f = [n, 2 * n, 4 * n, 8 * n, 16 * n]
    # (name, in_ch, out_ch) in the registration order of snunet.py:75-103 ; ("up", name, ch)
    return [
        ("block", "conv0_0", c, f[0]), ("block", "conv1_0", f[0], f[1]), ("up", "Up1_0", f[1]),
        ("block", "conv2_0", f[1], f[2]), ("up", "Up2_0", f[2]), ("block", "conv3_0", f[2], f[3]),
        ("up", "Up3_0", f[3]), ("block", "conv4_0", f[3], f[4]), ("up", "Up4_0", f[4]),
        ("block", "conv0_1", f[0] * 2 + f[1], f[0]), ("block", "conv1_1", f[1] * 2 + f[2], f[1]),
        ("up", "Up1_1", f[1]), ("block", "conv2_1", f[2] * 2 + f[3], f[2]), ("up", "Up2_1", f[2]),
        ("block", "conv3_1", f[3] * 2 + f[4], f[3]), ("up", "Up3_1", f[3]),
        ("block", "conv0_2", f[0] * 3 + f[1], f[0]), ("block", "conv1_2", f[1] * 3 + f[2], f[1]),
        ("up", "Up1_2", f[1]), ("block", "conv2_2", f[2] * 3 + f[3], f[2]), ("up", "Up2_2", f[2]),
        ("block", "conv0_3", f[0] * 4 + f[1], f[0]), ("block", "conv1_3", f[1] * 4 + f[2], f[1]),
        ("up", "Up1_3", f[1]), ("block", "conv0_4", f[0] * 5 + f[1], f[0]),
    ]


class _Holder(nn.Module):
    """Parameter/buffer container that only exists to reproduce the reference's key names."""


class SNUNet_ECAM(nn.Module):
    def __init__(self, in_channels, out_ch, base_channel=32, precision="bf16"):
        super().__init__()
        if out_ch != 3:
            raise _lib.KsmiError("SNUNet_ECAM (HIP): out_ch must be 3 (num_classes of the reference configs)")
        self.in_channels, self.out_ch, self.base_channel = in_channels, out_ch, base_channel
        self.precision = precision            # "bf16" (performance) | "fp32" (parity)
        # optional: BatchNorm statistics over the global batch under data parallelism (snunet_plan.SNUNetPlan.sync_bn; KSMI_SYNC_BN=1 or
        # configs["sync_bn"]); default = the reference's per-process BatchNorm
        import os as _os
        self.sync_bn = _os.environ.get("KSMI_SYNC_BN", "0") == "1"
        n = base_channel
        self._pspec, self._bspec, self._ispec = OrderedDict(), OrderedDict(), OrderedDict()
        for item in _blocks(n, in_channels):
            if item[0] == "block":
                _, name, cin, cout = item
                self._pspec[f"{name}.conv1.weight"] = (cout, cin, 3, 3)
                self._pspec[f"{name}.conv1.bias"] = (cout,)
                self._pspec[f"{name}.bn1.weight"] = (cout,)
                self._pspec[f"{name}.bn1.bias"] = (cout,)
                self._pspec[f"{name}.conv2.weight"] = (cout, cout, 3, 3)
                self._pspec[f"{name}.conv2.bias"] = (cout,)
                self._pspec[f"{name}.bn2.weight"] = (cout,)
                self._pspec[f"{name}.bn2.bias"] = (cout,)
                for bn in ("bn1", "bn2"):
                    self._bspec[f"{name}.{bn}.running_mean"] = (cout,)
                    self._bspec[f"{name}.{bn}.running_var"] = (cout,)
                    self._ispec[f"{name}.{bn}.num_batches_tracked"] = ()
            else:
                _, name, ch = item
                self._pspec[f"{name}.up.weight"] = (ch, ch, 2, 2)
                self._pspec[f"{name}.up.bias"] = (ch,)
        self._pspec["ca.fc1.weight"] = (4 * n // 16, 4 * n, 1, 1)
        self._pspec["ca.fc2.weight"] = (4 * n, 4 * n // 16, 1, 1)
        self._pspec["ca1.fc1.weight"] = (n // 4, n, 1, 1)
        self._pspec["ca1.fc2.weight"] = (n, n // 4, 1, 1)
        self._pspec["conv_final.weight"] = (out_ch, 4 * n, 1, 1)
        self._pspec["conv_final.bias"] = (out_ch,)
        self._build_arenas(torch.device("cpu"))
        self._init_parameters()
        self._plans = {}
        self._anchor = None
        self._raw_norm = None
        self._raw_version = 0          # bumped whenever the pipeline values change: part of the plan cache key

    # ------------------------------------------------------------------ raw-tile input
    def set_input_pipeline(self, mean=None, std=None, clamp=None):
        """Fold the Dataset's per-tile pipeline (dataset/Dataset.py:164-168 clamp to [0, clamp_input] + nan_to_num(clamp_input),
        :193-198 Normalize) into the image load of conv0_0.conv1: after this call forward() takes RAW tiles (NaNs included).
        `mean`, `std`: per input channel; `clamp`: a float (every channel) or per channel, a negative entry = no clamp for that
        channel (DEM: NaN -> mean).  set_input_pipeline() with no arguments returns to normalised inputs."""
        if mean is None:
            if self._raw_norm is not None:
                self._drop_raw_plans()
            self._raw_norm = None
            return self
        mean = [float(v) for v in mean]
        std = [float(v) for v in std]
        clamp = [float(clamp)] * len(mean) if not hasattr(clamp, "__len__") else [float(v) for v in clamp]
        if not (len(mean) == len(std) == len(clamp) == self.in_channels):
            raise ValueError(f"set_input_pipeline: {self.in_channels} input channels, got {len(mean)} means, {len(std)} stds, {len(clamp)} clamps")
        if any(v == 0.0 for v in std):
            raise ValueError("set_input_pipeline: a zero standard deviation")
        new = torch.tensor([mean, std, clamp], dtype=torch.float32)
        if self._raw_norm is None or not torch.equal(self._raw_norm.cpu(), new):       # (same values: the plans built on them stay valid)
            self._drop_raw_plans()                 # plans hold raw device pointers into the old tensor
            self._raw_norm = new
            self._raw_version += 1
        return self

    def _drop_raw_plans(self):
        for k in [k for k in self._plans if k[6] is not None]:
            del self._plans[k]

    def _raw_ptrs(self, dev):
        """(mean, std, clamp) device pointers for the first conv, or three nulls"""
        if self._raw_norm is None:
            return 0, 0, 0
        if self._raw_norm.device != dev:
            self._raw_norm = self._raw_norm.to(dev)
        base, step = self._raw_norm.data_ptr(), self.in_channels * 4
        return base, base + step, base + 2 * step

    # ------------------------------------------------------------------ arenas
    @staticmethod
    def _numel(shape):
        r = 1
        for s in shape:
            r *= s
        return r

    def _holder(self, path):
        mod = self
        for part in path:
            if not hasattr(mod, part):
                setattr(mod, part, _Holder())
            mod = getattr(mod, part)
        return mod

    def _build_arenas(self, device, old=None):
        """(Re)create the flat arenas on `device` and (re)register every parameter/buffer as a view."""
        def layout(spec, align):
            offs, o = OrderedDict(), 0
            for k, shp in spec.items():
                offs[k] = o
                o += -(-max(self._numel(shp), 1) // align) * align
            return offs, o
        self._poff, pn = layout(self._pspec, 4)        # 16-byte aligned views
        self._boff, bn = layout(self._bspec, 4)
        self._ioff, inn = layout(self._ispec, 1)
        self.flat_params = torch.zeros(pn, dtype=torch.float32, device=device)
        self.flat_grads = torch.zeros(pn, dtype=torch.float32, device=device)
        self.flat_buffers = torch.zeros(bn, dtype=torch.float32, device=device)
        self.flat_counters = torch.zeros(inn, dtype=torch.int64, device=device)
        order = []
        for item in _blocks(self.base_channel, self.in_channels):
            name = item[1]
            if item[0] == "block":
                order += [(f"{name}.conv1", ["weight", "bias"], []),
                          (f"{name}.bn1", ["weight", "bias"], ["running_mean", "running_var", "num_batches_tracked"]),
                          (f"{name}.conv2", ["weight", "bias"], []),
                          (f"{name}.bn2", ["weight", "bias"], ["running_mean", "running_var", "num_batches_tracked"])]
            else:
                order += [(f"{name}.up", ["weight", "bias"], [])]
        order += [("ca.fc1", ["weight"], []), ("ca.fc2", ["weight"], []), ("ca1.fc1", ["weight"], []),
                  ("ca1.fc2", ["weight"], []), ("conv_final", ["weight", "bias"], [])]
        for path, pnames, bnames in order:
            h = self._holder(path.split("."))
            for pn_ in pnames:
                key = f"{path}.{pn_}"
                shp = self._pspec[key]
                view = self.flat_params[self._poff[key]:self._poff[key] + self._numel(shp)].view(shp)
                if old is not None:
                    view.copy_(old[key])
                if pn_ in h._parameters and h._parameters[pn_] is not None:
                    h._parameters[pn_].data = view
                else:
                    h.register_parameter(pn_, nn.Parameter(view))
            for bn_ in bnames:
                key = f"{path}.{bn_}"
                if bn_ == "num_batches_tracked":
                    view = self.flat_counters[self._ioff[key]:self._ioff[key] + 1].view(())
                else:
                    shp = self._bspec[key]
                    view = self.flat_buffers[self._boff[key]:self._boff[key] + shp[0]]
                if old is not None:
                    view.copy_(old[key])
                h._buffers[bn_] = view
        self._arena_device = device
        self._plans = {}

    def _arena_ok(self):
        p0 = self.conv0_0.conv1.weight
        pl = self.conv_final.bias
        b0 = self.conv0_0.bn1.running_mean
        return (p0.data_ptr() == self.flat_params.data_ptr() + 4 * self._poff["conv0_0.conv1.weight"]
                and pl.data_ptr() == self.flat_params.data_ptr() + 4 * self._poff["conv_final.bias"]
                and b0.data_ptr() == self.flat_buffers.data_ptr() + 4 * self._boff["conv0_0.bn1.running_mean"]
                and p0.device == self.flat_params.device)

    def _ensure_arena(self):
        if not self._arena_ok():
            old = {k: v.detach().clone() for k, v in self.state_dict().items()}
            self._build_arenas(self.conv0_0.conv1.weight.device, old)

    def _init_parameters(self):
        """snunet.py:110-115: kaiming_normal_(fan_out, relu) on every nn.Conv2d weight, BN gamma=1
        beta=0; conv biases and ConvTranspose2d keep the PyTorch default init."""
        with torch.no_grad():
            for key, shp in self._pspec.items():
                p = self.flat_params[self._poff[key]:self._poff[key] + self._numel(shp)].view(shp)
                if ".up." in key:
                    if key.endswith("weight"):
                        nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                    else:
                        fan_in = shp[0] * 4 if False else self._pspec[key.replace("bias", "weight")][1] * 4
                        p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
                elif ".bn" in key:
                    p.fill_(1.0 if key.endswith("weight") else 0.0)
                elif key.endswith("weight"):
                    nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")
                else:
                    w = self._pspec[key.replace("bias", "weight")]
                    fan_in = w[1] * w[2] * w[3]
                    p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
            for key, shp in self._bspec.items():
                b = self.flat_buffers[self._boff[key]:self._boff[key] + shp[0]]
                b.fill_(1.0 if key.endswith("running_var") else 0.0)

    # views used by the plan
    def _p(self, key):
        shp = self._pspec[key]
        return self.flat_params[self._poff[key]:self._poff[key] + self._numel(shp)]

    def _g(self, key):
        shp = self._pspec[key]
        return self.flat_grads[self._poff[key]:self._poff[key] + self._numel(shp)]

    def _b(self, key):
        return self.flat_buffers[self._boff[key]:self._boff[key] + self._bspec[key][0]]

    def _c(self, key):
        return self.flat_counters[self._ioff[key]:self._ioff[key] + 1]

    def act_dtype(self):
        return torch.bfloat16 if self.precision == "bf16" else torch.float32

    # ------------------------------------------------------------------ forward
    def plan(self, B, H, W, training, with_backward, tail=0):
        self._ensure_arena()
        # the key carries a version counter, not id(tensor): _raw_ptrs() moves the tensor to the device (a new object), and a freed
        # tensor's id can be handed out again
        key = (B, H, W, self.act_dtype(), bool(training), bool(with_backward), None if self._raw_norm is None else self._raw_version, tail,
               bool(self.sync_bn))
        if key not in self._plans:
            from .snunet_plan import SNUNetPlan
            plan = SNUNetPlan(self, B, H, W, self.act_dtype(), training, with_backward, tail=tail, sync_bn=self.sync_bn)
            if self._raw_norm is not None:
                plan.keep.append(self._raw_norm)   # the plan's launches hold data_ptr()s into it
            self._plans[key] = plan
        return self._plans[key]

    def forward(self, xA, xB, dem=None):
        """xA, xB: the two dates [B,C,H,W].  dem (optional, [B,Cd,H,W]): channels shared by both dates; forward(xA, xB, dem) equals
        forward(cat(xA, dem), cat(xB, dem)) (the trainer's input assembly, change_detection_trainer.py:117-133) without the copies."""
        require_gpu(xA)
        tail = 0 if dem is None else dem.shape[1]
        if xA.shape != xB.shape or xA.dim() != 4 or xA.shape[1] + tail != self.in_channels or (dem is not None and (
                dem.dim() != 4 or dem.shape[0] != xA.shape[0] or dem.shape[2:] != xA.shape[2:])):
            raise ValueError(f"expected two [B,{self.in_channels - tail},H,W] tensors" + (f" and a [B,{tail},H,W] tail" if tail else "") +
                             f", got {tuple(xA.shape)} {tuple(xB.shape)}" + (f" {tuple(dem.shape)}" if tail else ""))
        B, _, H, W = xA.shape
        if H % 16 or W % 16:
            raise ValueError("H and W must be multiples of 16 (four 2x2 max-pools)")
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        plan = self.plan(B, H, W, self.training, want_grad, tail)
        xA = xA.contiguous().float()
        xB = xB.contiguous().float()
        dem = None if dem is None else dem.contiguous().float()
        if not want_grad:
            return plan.run_forward(xA, xB, dem).clone()
        if self._anchor is None or self._anchor.device != xA.device:
            self._anchor = torch.zeros(1, device=xA.device, requires_grad=True)
        return _SNUNetFn.apply(self._anchor, xA, xB, dem, self, plan)


class _SNUNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, xA, xB, dem, model, plan):
        from .arena import stamp_forward
        ctx.model, ctx.plan = model, plan
        ctx.gen = stamp_forward(plan)
        return plan.run_forward(xA, xB, dem).clone()

    @staticmethod
    def backward(ctx, dlogits):
        from .arena import check_forward_stamp
        model, plan = ctx.model, ctx.plan
        check_forward_stamp(plan, ctx.gen)
        params = list(model.parameters())
        if any(p.grad is not None for p in params):
            raise _lib.KsmiError("gradient accumulation across backward() calls is not supported by the HIP SNUNet: "
                                 "call optimizer.zero_grad(set_to_none=True) (the PyTorch default) before each step")
        plan.run_backward(dlogits.contiguous().float())
        for key in model._pspec:
            mod = model
            parts = key.split(".")
            for part in parts[:-1]:
                mod = getattr(mod, part)
            p = mod._parameters[parts[-1]]
            if p.requires_grad:
                p.grad = model._g(key).view(model._pspec[key])
        return None, None, None, None, None, None


