"""Flat-arena nn.Module base: every parameter is a view into ONE fp32 buffer (gradients into a
second one), registered under the reference's state-dict key names.  The optimiser step and the
data-parallel all-reduce are then single flat operations, and a model plan can hand raw device
pointers to the C-ABI.  (SNUNet_ECAM predates this class and carries its own copy with buffers.)
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib


class _Holder(nn.Module):
    """Container that only exists to reproduce the reference's key names."""


def _numel(shape):
    r = 1
    for s in shape:
        r *= s
    return r


class ArenaModule(nn.Module):
    def _setup_arena(self, pspec):
        """pspec: OrderedDict key -> shape, in the reference's registration order."""
        self._pspec = OrderedDict(pspec)
        self._plans = {}
        self._anchor = None
        self._build_arenas(torch.device("cpu"))

    def _holder(self, path):
        mod = self
        for part in path:
            if part not in mod._modules:
                mod.add_module(part, _Holder())
            mod = mod._modules[part]
        return mod

    def _build_arenas(self, device, old=None):
        offs, o = OrderedDict(), 0
        for k, shp in self._pspec.items():
            offs[k] = o
            o += -(-max(_numel(shp), 1) // 4) * 4            # 16-byte aligned views
        self._poff = offs
        self.flat_params = torch.zeros(o, dtype=torch.float32, device=device)
        self.flat_grads = torch.zeros(o, dtype=torch.float32, device=device)
        for key, shp in self._pspec.items():
            parts = key.split(".")
            h = self._holder(parts[:-1])
            view = self.flat_params[offs[key]:offs[key] + _numel(shp)].view(shp)
            if old is not None:
                view.copy_(old[key])
            if parts[-1] in h._parameters and h._parameters[parts[-1]] is not None:
                h._parameters[parts[-1]].data = view
            else:
                h.register_parameter(parts[-1], nn.Parameter(view))
        self._plans = {}

    def _param_obj(self, key):
        parts = key.split(".")
        mod = self
        for part in parts[:-1]:
            mod = mod._modules[part]
        return mod._parameters[parts[-1]]

    def _arena_ok(self):
        keys = list(self._pspec)
        for key in (keys[0], keys[-1]):
            p = self._param_obj(key)
            if p.device != self.flat_params.device or p.data_ptr() != self.flat_params.data_ptr() + 4 * self._poff[key]:
                return False
        return True

    def _ensure_arena(self):
        if not self._arena_ok():
            old = {k: self._param_obj(k).detach().clone() for k in self._pspec}
            self._build_arenas(self._param_obj(next(iter(self._pspec))).device, old)

    def _p(self, key):
        return self.flat_params[self._poff[key]:self._poff[key] + _numel(self._pspec[key])]

    def _g(self, key):
        return self.flat_grads[self._poff[key]:self._poff[key] + _numel(self._pspec[key])]

    def act_dtype(self):
        return torch.bfloat16 if self.precision == "bf16" else torch.float32

    def _attach_grads(self):
        for key, shp in self._pspec.items():
            p = self._param_obj(key)
            if p.requires_grad:
                p.grad = self._g(key).view(shp)

    def _check_no_grads(self):
        if any(p.grad is not None for p in self.parameters()):
            raise _lib.KsmiError("gradient accumulation across backward() calls is not supported by the HIP plan: "
                                 "call optimizer.zero_grad(set_to_none=True) (the PyTorch default) before each step")


class PlanFn(torch.autograd.Function):
    """Whole-model autograd node: forward replays plan.fwd, backward replays plan.bwd and attaches the
    arena-backed .grad views (anchor = dummy leaf that makes autograd call us)."""

    @staticmethod
    def forward(ctx, anchor, model, plan, *inputs):
        ctx.model, ctx.plan = model, plan
        return plan.run_forward(*inputs).clone()

    @staticmethod
    def backward(ctx, dout):
        model, plan = ctx.model, ctx.plan
        model._check_no_grads()
        plan.run_backward(dout.contiguous().float())
        model._attach_grads()
        return (None,) * (3 + len(plan.input_names))
