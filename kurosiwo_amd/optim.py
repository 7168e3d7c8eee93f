"""Fused optimisers over the flat parameter arena (row O1):
torch.optim.Adam(lr) / SGD(momentum, weight_decay) as constructed at
/root/reference/training/change_detection_trainer.py:45-66.

One HIP kernel updates the whole model (12 M parameters = one 48 MB pass) instead of a
236-tensor foreach loop.  The step counter lives on the device so a captured HIP graph
can be replayed.
"""
import torch

from . import _lib
from .runtime import stream_ptr


def _arena_of(params):
    """(base_ptr, numel) if the tensors tile one contiguous fp32 buffer in order, else None."""
    params = [p for p in params]
    if not params:
        return None
    base = params[0].data_ptr()
    end = base
    for p in params:
        if p.dtype != torch.float32 or not p.is_contiguous() or p.data_ptr() < end or p.data_ptr() - end > 12:
            return None
        end = p.data_ptr() + 4 * p.numel()
    return base, (end - base) // 4


class _FlatOptimizer(torch.optim.Optimizer):
    def _flat_state(self, group, names):
        st = self.state.setdefault("flat%d" % id(group), {})
        ps = group["params"]
        ar = _arena_of(ps)
        if ar is None:
            raise _lib.KsmiError("fused optimiser: parameters must be views of one flat fp32 arena "
                                 "(pass model.parameters() of a kurosiwo_amd model)")
        if "n" not in st or st["n"] != ar[1] or st["base"] != ar[0]:
            dev = ps[0].device
            st["base"], st["n"] = ar
            for nm in names:
                st[nm] = torch.zeros(ar[1], dtype=torch.float32, device=dev)
            st["step"] = torch.zeros(1, dtype=torch.int64, device=dev)
        gr = _arena_of([p.grad for p in ps]) if all(p.grad is not None for p in ps) else None
        if gr is None or gr[1] != ar[1]:
            raise _lib.KsmiError("fused optimiser: gradients are not a flat arena (run backward through the HIP model first)")
        return st, ar, gr


class FusedAdam(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = _lib.load()
        for g in self.param_groups:
            st, ar, gr = self._flat_state(g, ("exp_avg", "exp_avg_sq"))
            _lib.check(lib.ksmi_adam_step(ar[0], gr[0], st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), ar[1],
                                          st["step"].data_ptr(), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                          g["weight_decay"], g["grad_scale"], stream_ptr()), "adam_step")
        return loss


class FusedSGD(_FlatOptimizer):
    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, grad_scale=grad_scale))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = _lib.load()
        for g in self.param_groups:
            st, ar, gr = self._flat_state(g, ("momentum_buffer",))
            _lib.check(lib.ksmi_sgd_step(ar[0], gr[0], st["momentum_buffer"].data_ptr(), ar[1], st["step"].data_ptr(),
                                         g["lr"], g["momentum"], g["weight_decay"], g["grad_scale"], stream_ptr()), "sgd_step")
        return loss
