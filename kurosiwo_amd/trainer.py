"""Fused change-detection train step on the HIP plan: the body of the reference's hot loop
/root/reference/training/change_detection_trainer.py:135-180 (zero_grad -> model(*inputs) ->
criterion -> backward -> optimizer.step) as one launch sequence on one stream, with the
data-parallel gradient reduction of kurosiwo_amd/dp.py overlapped with backward.
"""
import torch
import torch.distributed as dist

from . import _lib
from .dp import BucketedAllReduce, make_buckets
from .runtime import stream_ptr


class CDTrainStep:
    def __init__(self, model, B, H, W, loss_function="ce+dice", class_weights=(1.0, 1.0, 1.0), lr=1e-3,
                 betas=(0.9, 0.999), eps=1e-8, optimizer="adam", momentum=0.0, weight_decay=0.0,
                 bucket_mb=25.0, group=None):
        if loss_function not in ("ce+dice", "cross_entropy"):
            raise NotImplementedError(loss_function)
        self.model = model
        self.lib = _lib.load()
        self.plan = model.plan(B, H, W, True, True)
        dev = self.plan.dev
        self.B, self.HW = B, H * W
        self.with_dice = 1 if loss_function == "ce+dice" else 0
        self.cw = torch.tensor(list(class_weights), dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(3, dtype=torch.float32, device=dev)
        self.loss_ws = torch.empty(self.lib.ksmi_loss_workspace(B, self.HW), dtype=torch.uint8, device=dev)
        self.labels = torch.empty((B, H, W), dtype=torch.int64, device=dev)
        n = model.flat_params.numel()
        self.opt_kind = optimizer
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev) if optimizer == "adam" else None
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.hp = dict(lr=lr, betas=betas, eps=eps, momentum=momentum, weight_decay=weight_decay)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        ready = {k: self.plan.param_ready.get(k, -1) for k in model._poff}
        numels = {k: model._numel(s) for k, s in model._pspec.items()}
        buckets = make_buckets(ready, model._poff, numels, n, int(bucket_mb * 1e6 / 4))
        self.reducer = BucketedAllReduce(model.flat_grads, buckets, group)
        self.timer = None          # optional kernel timer (bench.py)

    def set_batch(self, xA, xB, labels):
        self.plan.xA.copy_(xA, non_blocking=True)
        self.plan.xB.copy_(xB, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)

    def run(self):
        """One train step on the batch currently resident in the plan's input buffers."""
        p, lib, st = self.plan, self.lib, stream_ptr()
        t = self.timer
        tt = (lambda k: t is not None and t.wants(k))
        p.packs.run(t)
        p.fwd.run(t)
        B, HW = self.B, self.HW
        tk = tt("ce_dice_forward")
        if tk: t.begin("ce_dice_forward")
        _lib.check(lib.ksmi_ce_dice_forward(p.logits.data_ptr(), self.labels.data_ptr(), self.cw.data_ptr(), self.with_dice,
                                            self.loss_out.data_ptr(), self.loss_ws.data_ptr(), B, HW, 3, st), "ce_dice_forward")
        if tk: t.end()
        tk = tt("ce_dice_backward")
        if tk: t.begin("ce_dice_backward")
        _lib.check(lib.ksmi_ce_dice_backward(p.logits.data_ptr(), self.labels.data_ptr(), self.cw.data_ptr(), self.with_dice,
                                             self.loss_ws.data_ptr(), None, p.dlogits.data_ptr(), B, HW, 3, st), "ce_dice_backward")
        if tk: t.end()
        p.bwd.run(t, self.reducer.after_launch)
        self.reducer.wait()
        gs = 1.0 / self.world
        mf = self.model
        tk = tt("optimizer")
        if tk: t.begin("optimizer")
        if self.opt_kind == "adam":
            _lib.check(lib.ksmi_adam_step(mf.flat_params.data_ptr(), mf.flat_grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                          mf.flat_params.numel(), self.step_count.data_ptr(), self.hp["lr"], self.hp["betas"][0],
                                          self.hp["betas"][1], self.hp["eps"], self.hp["weight_decay"], gs, st), "adam_step")
        else:
            _lib.check(lib.ksmi_sgd_step(mf.flat_params.data_ptr(), mf.flat_grads.data_ptr(), self.m.data_ptr(),
                                         mf.flat_params.numel(), self.step_count.data_ptr(), self.hp["lr"], self.hp["momentum"],
                                         self.hp["weight_decay"], gs, st), "sgd_step")
        if tk: t.end()

    def step(self, xA, xB, labels):
        self.set_batch(xA, xB, labels)
        self.run()
        return self.loss_out      # device tensor [total, ce, dice]; no host sync here
