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


def _as_i32(words):
    """two unsigned 32-bit words as the int32 tensor torch can hold"""
    return torch.tensor([v - (1 << 32) if v >= (1 << 31) else v for v in words], dtype=torch.int32)


class ArenaModule(nn.Module):
    def _setup_arena(self, pspec, bspec=None, ispec=None):
        """pspec: OrderedDict key -> shape, in the reference's registration order; bspec: fp32 buffers (BatchNorm running
        statistics); ispec: int64 counters (num_batches_tracked)."""
        self._pspec = OrderedDict(pspec)
        self._bspec = OrderedDict(bspec or {})
        self._ispec = OrderedDict(ispec or {})
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
            o += -(-max(_numel(shp), 1) // 8) * 8            # 32-byte aligned views (16-byte aligned rows in the bf16 mirror)
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
        boffs, o = OrderedDict(), 0
        for k, shp in self._bspec.items():
            boffs[k] = o
            o += -(-max(_numel(shp), 1) // 4) * 4
        self._boff = boffs
        self.flat_buffers = torch.zeros(max(o, 1), dtype=torch.float32, device=device)
        self._ioff = OrderedDict((k, i) for i, k in enumerate(self._ispec))
        self.flat_counters = torch.zeros(max(len(self._ispec), 1), dtype=torch.int64, device=device)
        for key in list(self._bspec) + list(self._ispec):
            parts = key.split(".")
            h = self._holder(parts[:-1])
            view = self._b(key).view(self._bspec[key]) if key in self._bspec else self._c(key).view(())
            if old is not None:
                view.copy_(old[key])
            h._buffers[parts[-1]] = view
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

    def _b(self, key):
        return self.flat_buffers[self._boff[key]:self._boff[key] + _numel(self._bspec[key])]

    def _c(self, key):
        return self.flat_counters[self._ioff[key]:self._ioff[key] + 1]

    def _buffer_obj(self, key):
        parts = key.split(".")
        mod = self
        for part in parts[:-1]:
            mod = mod._modules[part]
        return mod._buffers[parts[-1]]

    def _ensure_arena(self):
        ok = self._arena_ok()
        if ok and self._bspec:
            k0 = next(iter(self._bspec))
            b = self._buffer_obj(k0)
            ok = b.device == self.flat_buffers.device and b.data_ptr() == self.flat_buffers.data_ptr() + 4 * self._boff[k0]
        if not ok:
            old = {k: self._param_obj(k).detach().clone() for k in self._pspec}
            old.update({k: self._buffer_obj(k).detach().clone() for k in list(self._bspec) + list(self._ispec)})
            self._build_arenas(self._param_obj(next(iter(self._pspec))).device, old)

    def _p(self, key):
        return self.flat_params[self._poff[key]:self._poff[key] + _numel(self._pspec[key])]

    def _g(self, key):
        return self.flat_grads[self._poff[key]:self._poff[key] + _numel(self._pspec[key])]

    # ---- random stream of the stochastic layers (nn.Dropout / DropPath / nn.Dropout2d of the reference models; csrc/common.h)
    @staticmethod
    def _fold_rank(seed):
        # every rank runs with the same torch seed (same loader shuffle): fold the rank into the stream so that sample i of
        # rank 0 and sample i of rank 1 draw different Dropout / DropPath masks (rank 0's folded seed is the base seed itself)
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        return (int(seed) ^ (0x9E3779B1 * rank)) & 0xFFFFFFFF

    def manual_seed(self, seed, step=0, fold_rank=False):
        """pin the counter-based stream: the next training forward uses (seed, step + 1).  fold_rank: `seed` is a BASE seed (what a
        checkpoint written by rank 0 holds) and this rank's share of the stream is derived from it, as the default seed is"""
        if fold_rank:
            seed = self._fold_rank(seed)
        self._rng_init = (int(seed) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF)
        if getattr(self, "_rng_state", None) is not None:
            self._rng_state.copy_(_as_i32(self._rng_init))
        return self

    def rng_state(self):
        """device words {seed, step} read by the dropout kernels (default seed: torch.initial_seed(), so torch.manual_seed steers it)"""
        self._ensure_arena()
        dev = self.flat_params.device
        if getattr(self, "_rng_state", None) is None or self._rng_state.device != dev:
            if getattr(self, "_rng_init", None) is None:
                self._rng_init = (self._fold_rank(torch.initial_seed()), 0)
            self._rng_state = _as_i32(self._rng_init).to(dev)
        return self._rng_state

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


def stamp_forward(plan):
    """A plan owns ONE set of activation buffers: every grad-enabled forward gets a generation number and backward() refuses to
    run on buffers a later forward of the same (shape, mode) plan has overwritten."""
    plan._generation = getattr(plan, "_generation", 0) + 1
    return plan._generation


def check_forward_stamp(plan, gen):
    if getattr(plan, "_generation", gen) != gen:
        raise _lib.KsmiError("backward() of a forward whose activations were overwritten: the HIP plan keeps one set of activation buffers "
                             "per (batch shape, mode), so call backward() before the next grad-enabled forward of the same shape")


class PlanFn(torch.autograd.Function):
    """Whole-model autograd node: forward replays plan.fwd, backward replays plan.bwd and attaches the
    arena-backed .grad views (anchor = dummy leaf that makes autograd call us)."""

    @staticmethod
    def forward(ctx, anchor, model, plan, *inputs):
        ctx.model, ctx.plan = model, plan
        ctx.gen = stamp_forward(plan)
        return plan.run_forward(*inputs).clone()

    @staticmethod
    def backward(ctx, dout):
        model, plan = ctx.model, ctx.plan
        check_forward_stamp(plan, ctx.gen)
        model._check_no_grads()
        plan.run_backward(dout.contiguous().float())
        model._attach_grads()
        return (None,) * (3 + len(plan.input_names))
