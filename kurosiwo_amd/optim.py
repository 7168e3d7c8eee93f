"""Fused optimisers over the flat parameter arena (row O1):
torch.optim.Adam(lr) / SGD(momentum, weight_decay) as constructed at
/root/reference/training/change_detection_trainer.py:45-66.

One HIP kernel updates the whole model (12 M parameters = one 48 MB pass) instead of a
236-tensor foreach loop.  The step counter lives on the device so a captured HIP graph
can be replayed.  They are real torch.optim.Optimizer subclasses (param_groups, lr schedulers,
state_dict work); `step()` serves the autograd path, `step_arena()` the fused train step.
"""
import torch

from . import _lib
from .runtime import stream_ptr


# Generation of every parameter arena a fused optimiser has written (keyed by the arena's base pointer): EVERY step_arena call bumps
# it, mirror or not, whichever plan or train step drove it.  The plans' bf16 operand copies (plan_base.mirror_written /
# _mirror_is_fresh) are fresh only while the generation still has the value their own mirrored step recorded: the optimiser writes the
# parameters through raw pointers, which torch's version counter never sees, so a second plan of the same model that steps in between
# would otherwise leave the first plan's copy one step stale.
_ARENA_GEN = {}


def arena_generation(p_ptr):
    return _ARENA_GEN.get(int(p_ptr), 0)


def _bump_generation(p_ptr):
    _ARENA_GEN[int(p_ptr)] = _ARENA_GEN.get(int(p_ptr), 0) + 1


def _arena_of(params):
    """(base_ptr, numel) of the flat fp32 arena the tensors are views of, in order, else None.  The arena is taken from the
    views' common storage (model.flat_params / flat_grads), so its numel is the model's own (SNUNet aligns its views to 4
    floats, ArenaModule to 8: the padding between views belongs to the arena and the optimiser state must cover it, or the
    autograd path (`opt.step()`) and the fused path (`step_arena`) would disagree on the state size and reset it)."""
    params = list(params)
    if not params or params[0].dtype != torch.float32:
        return None
    st = params[0].untyped_storage()                       # (nn.Parameter(view) drops ._base; the storage is the arena)
    lo, nbytes = st.data_ptr(), st.nbytes()
    if nbytes % 4:
        return None
    end = lo
    for p in params:
        if (p.dtype != torch.float32 or not p.is_contiguous() or p.untyped_storage().data_ptr() != lo or p.data_ptr() < end
                or p.data_ptr() + 4 * p.numel() > lo + nbytes):
            return None
        end = p.data_ptr() + 4 * p.numel()
    return lo, nbytes // 4


class _FlatOptimizer(torch.optim.Optimizer):
    _names = ()

    def flat_state(self, n, device):
        st = self.state.setdefault("flat", {})
        if st.get("n") != n:
            st["n"] = n
            for nm in self._names:
                st[nm] = torch.zeros(n, dtype=torch.float32, device=device)
            st["step"] = torch.zeros(1, dtype=torch.int64, device=device)
        return st

    def _arenas(self):
        if len(self.param_groups) != 1:
            raise _lib.KsmiError("fused optimiser: exactly one param group is supported")
        ps = self.param_groups[0]["params"]
        ar = _arena_of(ps)
        if ar is None:
            raise _lib.KsmiError("fused optimiser: parameters must be views of one flat fp32 arena "
                                 "(pass model.parameters() of a kurosiwo_amd model)")
        gr = _arena_of([p.grad for p in ps]) if all(p.grad is not None for p in ps) else None
        if gr is None or gr[1] != ar[1]:
            raise _lib.KsmiError("fused optimiser: gradients are not a flat arena (run backward through the HIP model first)")
        return ar, gr, ps[0].device

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        ar, gr, dev = self._arenas()
        self.step_arena(ar[0], gr[0], ar[1], dev)
        return loss


class FusedAdam(_FlatOptimizer):
    _names = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    _decoupled = 0

    def step_arena(self, p_ptr, g_ptr, n, device, grad_scale=1.0, mirror=None):
        """mirror: device pointer of a bf16 copy of the parameter arena that the step refreshes as it writes the parameters
        (ksmi_adam_step_mirror; the plans' operand copy for the token GEMMs).  Returns True when it did."""
        g = self.param_groups[0]
        st = self.flat_state(n, device)
        _bump_generation(p_ptr)
        if mirror:
            _lib.check(_lib.load().ksmi_adam_step_mirror(p_ptr, g_ptr, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n,
                                                         st["step"].data_ptr(), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                                         g["weight_decay"], grad_scale, self._decoupled, mirror, stream_ptr()), "adam_step_mirror")
            return True
        self._step_plain(p_ptr, g_ptr, n, st, g, grad_scale)
        return False

    def _step_plain(self, p_ptr, g_ptr, n, st, g, grad_scale):
        _lib.check(_lib.load().ksmi_adam_step(p_ptr, g_ptr, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n,
                                              st["step"].data_ptr(), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                              g["weight_decay"], grad_scale, stream_ptr()), "adam_step")


class FusedAdamW(FusedAdam):
    """torch.optim.AdamW on the flat arena (training/change_detection_trainer.py:55-60: betas, weight_decay from the method json)"""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    _decoupled = 1

    def _step_plain(self, p_ptr, g_ptr, n, st, g, grad_scale):
        _lib.check(_lib.load().ksmi_adamw_step(p_ptr, g_ptr, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n,
                                               st["step"].data_ptr(), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                               g["weight_decay"], grad_scale, stream_ptr()), "adamw_step")


class FusedSGD(_FlatOptimizer):
    _names = ("momentum_buffer",)

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    def step_arena(self, p_ptr, g_ptr, n, device, grad_scale=1.0):
        g = self.param_groups[0]
        st = self.flat_state(n, device)
        _bump_generation(p_ptr)
        _lib.check(_lib.load().ksmi_sgd_step(p_ptr, g_ptr, st["momentum_buffer"].data_ptr(), n, st["step"].data_ptr(),
                                             g["lr"], g["momentum"], g["weight_decay"], grad_scale, stream_ptr()), "sgd_step")
