"""Fused change-detection train step on the HIP plan: the body of the reference's hot loop
/root/reference/training/change_detection_trainer.py:135-180 (zero_grad -> model(*inputs) ->
criterion -> backward -> optimizer.step) as one launch sequence on one stream, with the
data-parallel gradient reduction of kurosiwo_amd/dp.py overlapped with backward.
"""
import os

import torch
import torch.distributed as dist

from . import _lib
from .dp import BucketedAllReduce, default_grad_dtype, make_buckets
from .optim import FusedAdam
from .runtime import stream_ptr


def _optimizer_step(step, mf, plan):
    """optimizer.step on the flat arenas; an Adam-family optimiser also refreshes the plan's bf16 operand copy of the parameters (the
    next forward then skips its cast pass: plan_base.mirror_written)"""
    opt = step.optimizer
    args = (mf.flat_params.data_ptr(), mf.flat_grads.data_ptr(), mf.flat_params.numel(), plan.dev, 1.0 / step.world)
    mirror = plan.mirror_ptr() if hasattr(plan, "mirror_ptr") and hasattr(opt, "_decoupled") else None
    if mirror and opt.step_arena(*args, mirror=mirror):
        plan.mirror_written()
    elif not mirror:
        opt.step_arena(*args)


class _PlanTrainStep:
    """zero_grad -> forward -> criterion -> backward (+ bucketed all-reduce) -> optimizer.step over a model plan."""

    def __init__(self, model, plan, B, H, W, loss_function="ce+dice", class_weights=(1.0, 1.0, 1.0), optimizer=None, lr=1e-3,
                 bucket_mb=8.0, group=None, graph=False, overlap_wgrad=True, overlap_lanes=True, grad_dtype=None, dp_mode=None):
        if loss_function not in ("ce+dice", "cross_entropy"):
            raise NotImplementedError(loss_function)
        self.model = model
        self.lib = _lib.load()
        self.plan = plan
        dev = self.plan.dev
        self.B, self.HW = B, H * W
        self.with_dice = 1 if loss_function == "ce+dice" else 0
        self.cw = torch.tensor(list(class_weights), dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(3, dtype=torch.float32, device=dev)
        self.loss_ws = torch.empty(self.lib.ksmi_loss_workspace(B, self.HW), dtype=torch.uint8, device=dev)
        self.labels = torch.empty((B, H, W), dtype=torch.int64, device=dev)
        self.optimizer = optimizer if optimizer is not None else FusedAdam(model.parameters(), lr=lr)
        n = model.flat_params.numel()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        ready = {k: self.plan.param_ready.get(k, -1) for k in model._poff}
        if os.environ.get("KSMI_DP_BUCKET_MB"):                        # A/B knob: gradient bucket size of the data-parallel all-reduce
            bucket_mb = float(os.environ["KSMI_DP_BUCKET_MB"])
        buckets = make_buckets(ready, model._poff, None, n, int(bucket_mb * 1e6 / 4))
        self.grad_dtype = grad_dtype or default_grad_dtype(n)          # wire format of the gradient buckets (dp.py)
        self.reducer = BucketedAllReduce(model.flat_grads, buckets, group, self.grad_dtype if self.world > 1 or os.environ.get("KSMI_DP_FORCE") else "fp32",
                                         mode=dp_mode)
        # a ready bucket may hold gradients written on any stream of the step: the reducer's issue stream waits for an event on each
        # of them (no compute stream is made to wait for another one because a bucket became ready)
        self.reducer.writer_streams = self._writer_streams
        self.timer = None          # optional kernel timer (bench.py)
        self.use_graph = bool(graph) and self.world == 1     # replay the step as one captured HIP graph (configs["hip_graph"])
        self.overlap_wgrad = bool(overlap_wgrad) and os.environ.get("KSMI_OVERLAP_WGRAD", "1") != "0"
        self.overlap_lanes = bool(overlap_lanes) and os.environ.get("KSMI_OVERLAP_LANES", "1") != "0"

    def _set_inputs(self, *inputs):
        raise NotImplementedError

    # ---- streams of the step (snunet_plan.StepStreams): the weight-gradient launches of the backward pass run on a side stream next to
    # the bandwidth-bound BatchNorm launches of the critical path (overlap_wgrad; plans with side_wgrad), and the deeper decoder blocks
    # of SNUNet on a second compute lane next to the level-0 column (overlap_lanes; plans with two_lanes).  Switches: the constructor
    # arguments / configs["overlap_wgrad"], configs["overlap_lanes"] / KSMI_OVERLAP_WGRAD=0, KSMI_OVERLAP_LANES=0.
    _ss = None

    def _streams(self):
        side = self.overlap_wgrad and (getattr(self.plan, "side_wgrad", False) or getattr(self.plan, "side_tokens", False))
        lanes = self.overlap_lanes and getattr(self.plan, "two_lanes", False)
        if not (side or lanes):
            return None
        if self._ss is None or (self._ss.lanes, self._ss.use_side) != (lanes, side):
            from .snunet_plan import StepStreams
            self._ss = StepStreams(self.plan.dev, lanes=lanes, side=side)
        return self._ss

    def _writer_streams(self):
        ss = self._ss
        if ss is None or ss.main is None:
            return [torch.cuda.current_stream()]
        return ss.all_streams()

    def _after_launch(self, idx):
        """bucket hook of the backward launch list (the reducer orders its collective behind every stream of the step itself)"""
        self.reducer.after_launch(idx)

    def set_batch(self, *args):
        self._set_inputs(*args[:-1])
        self.labels.copy_(args[-1], non_blocking=True)

    def _timed(self, kind, fn):
        t = self.timer
        on = t is not None and t.wants(kind)
        if on:
            t.begin(kind)
        fn()
        if on:
            t.end()

    # ---- HIP graph: the step is a static launch list on one stream (no allocation, no host sync, device-side step counters and
    # random-stream state), so it can be captured once and replayed -- the CPU then issues ONE graph launch per step instead of
    # several hundred ctypes calls.  Single-GPU only (the RCCL bucket hooks stay eager); the learning rate is baked at capture time
    # and a change re-captures.
    _graph = None

    def capture_graph(self):
        self.plan._mirror_off = True                 # (a replayed graph cannot re-decide whether the cast pass is needed: keep it in)
        if self.world > 1:
            raise _lib.KsmiError("graph capture of the train step is single-GPU (the bucketed all-reduce hooks run eagerly)")
        if self.timer is not None:
            raise _lib.KsmiError("graph capture with a kernel timer attached")
        self._graph = None
        self._run_eager()                            # warm-up: lazy attribute / symbol look-ups happen outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._run_eager()
        self._graph, self._graph_hp = g, self._baked_hyper()
        return self

    def _baked_hyper(self):
        """every optimiser scalar a captured step bakes into its kernel arguments (a change of any of them re-captures)"""
        keys = ("lr", "betas", "eps", "weight_decay", "momentum", "dampening", "nesterov", "amsgrad")
        return tuple(tuple((k, g[k]) for k in keys if k in g) for g in self.optimizer.param_groups) + (self.world,)

    def run(self):
        """One train step on the batch currently resident in the plan's input buffers."""
        if self._graph is not None and self.timer is None:
            if self._baked_hyper() != self._graph_hp:
                self._recaptures = getattr(self, "_recaptures", 0) + 1
                if self._recaptures == 8:
                    import warnings
                    warnings.warn("hip_graph: the optimiser hyper-parameters changed in 8 steps (a per-iteration LR schedule?); every "
                                  "change costs one eager step + a capture -- run such schedules without hip_graph")
                self.capture_graph()
                return
            self._graph.replay()
            return
        if getattr(self, "use_graph", False) and self.timer is None:
            self.capture_graph()                     # first step: one eager step + the capture
            return
        self._run_eager()

    def _run_eager(self):
        p, lib, st = self.plan, self.lib, stream_ptr()
        t = self.timer
        p.packs.run(t)
        p.fwd.run(t, None, self._streams())
        B, HW = self.B, self.HW
        self._timed("ce_dice_forward", lambda: _lib.check(lib.ksmi_ce_dice_forward(
            p.logits.data_ptr(), self.labels.data_ptr(), self.cw.data_ptr(), self.with_dice, self.loss_out.data_ptr(),
            self.loss_ws.data_ptr(), B, HW, 3, st), "ce_dice_forward"))
        self._timed("ce_dice_backward", lambda: _lib.check(lib.ksmi_ce_dice_backward(
            p.logits.data_ptr(), self.labels.data_ptr(), self.cw.data_ptr(), self.with_dice, self.loss_ws.data_ptr(), None,
            p.dlogits.data_ptr(), B, HW, 3, st), "ce_dice_backward"))
        ss = self._streams()
        p.bwd.run(t, self._after_launch if ss is not None else self.reducer.after_launch, ss, hook_at=self.reducer.hook_indices())
        if ss is not None:
            ss.end()
        self.reducer.wait()
        mf = self.model
        self._timed("optimizer", lambda: _optimizer_step(self, mf, p))

    def step(self, *args):
        self.set_batch(*args)
        self.run()
        return self.loss_out      # device tensor [total, ce, dice]; no host sync here


class CDTrainStep(_PlanTrainStep):
    """change_detection_trainer.py:135-180 on SNUNet_ECAM: step(xA, xB, labels)."""

    def __init__(self, model, B, H, W, loss_function="ce+dice", class_weights=(1.0, 1.0, 1.0), tail=0, **kw):
        """tail > 0 (SNUNet_ECAM only): step(xA, xB, dem, labels) with `tail` channels shared by both dates read straight from their own
        buffer by the first convolution (snunet.SNUNet_ECAM.forward's `dem`)"""
        plan = model.plan(B, H, W, True, True, tail) if tail else model.plan(B, H, W, True, True)
        super().__init__(model, plan, B, H, W, loss_function, class_weights, **kw)

    def _set_inputs(self, xA, xB, tail=None):
        self.plan.xA.copy_(xA, non_blocking=True)
        self.plan.xB.copy_(xB, non_blocking=True)
        if (tail is None) != (getattr(self.plan, "xtail", None) is None):
            raise _lib.KsmiError("CDTrainStep: built with / without tail channels, called the other way")
        if tail is not None:
            self.plan.xtail.copy_(tail, non_blocking=True)


class SegTrainStep(_PlanTrainStep):
    """segmentation_trainer.py:54-171 on FinetunerSegmentation (FloodViT): step(x, labels) with x the channel concat
    [post, (dem), pre1, pre2] (:107-147); default criterion = create_loss 'cross_entropy' with class weights."""

    def __init__(self, model, B, loss_function="cross_entropy", class_weights=(1.0, 1.0, 1.0), **kw):
        if hasattr(model, "hp"):                      # FloodViT: fixed image size
            ih, iw = model.hp["image_size"]
            plan = model.plan(B, True, True)
        else:                                         # Unet(resnet18): plan per (B, H, W)
            ih, iw = kw.pop("image_size", (224, 224))
            plan = model.plan(B, ih, iw, True, True)
        super().__init__(model, plan, B, ih, iw, loss_function, class_weights, **kw)

    def _set_inputs(self, x):
        self.plan.x.copy_(x, non_blocking=True)


class MAETrainStep:
    """training/train_mae.py:62-122 on the MAE plan: step(image) = zero_grad -> mae(image) (fresh random permutation,
    models/mae.py:73) -> backward (+ bucketed all-reduce) -> Adam, as one launch sequence.  loss_out[0] = reconstruction loss."""

    def __init__(self, model, B, optimizer=None, lr=1e-5, bucket_mb=32.0, group=None, loss_scale=1.0, grad_dtype=None, dp_mode=None):
        self.model, self.B = model, B
        self.lib = _lib.load()
        self.plan = model.plan(B, True)
        self.optimizer = optimizer if optimizer is not None else FusedAdam(model.parameters(), lr=lr)
        n = model.flat_params.numel()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        ready = {k: self.plan.param_ready.get(k, -1) for k in model._poff}
        self.grad_dtype = grad_dtype or default_grad_dtype(n)
        self.reducer = BucketedAllReduce(model.flat_grads, make_buckets(ready, model._poff, None, n, int(bucket_mb * 1e6 / 4)), group,
                                         self.grad_dtype if self.world > 1 or os.environ.get("KSMI_DP_FORCE") else "fp32", mode=dp_mode)
        self.reducer.writer_streams = lambda: ([torch.cuda.current_stream()] if self._ss is None or self._ss.main is None
                                               else self._ss.all_streams())
        self.loss_out = self.plan.loss
        self.plan.dloss.fill_(loss_scale)
        self.timer = None

    # the nn.Linear weight gradients of the transformer layers on a side stream (plan_base.PlanBase.side_tokens; KSMI_OVERLAP_WGRAD=0: off)
    _ss = None

    def _streams(self):
        if not (getattr(self.plan, "side_tokens", False) and os.environ.get("KSMI_OVERLAP_WGRAD", "1") != "0"):
            return None
        if self._ss is None:
            from .snunet_plan import StepStreams
            self._ss = StepStreams(self.plan.dev, lanes=False, side=True)
        return self._ss

    def _after_launch(self, idx):
        self.reducer.after_launch(idx)

    def set_batch(self, image, rand_indices=None):
        self.plan.x.copy_(image, non_blocking=True)
        if rand_indices is not None:
            self.plan.idx.copy_(rand_indices, non_blocking=True)
        self._fixed_idx = rand_indices is not None

    def run(self):
        p, t = self.plan, self.timer
        if not getattr(self, "_fixed_idx", False):                 # mae.py:73
            p.idx.copy_(torch.rand(self.B, p.N, device=p.dev).argsort(dim=-1))
        p.packs.run(t)
        p.fwd.run(t)
        ss = self._streams()
        p.bwd.run(t, self._after_launch if ss is not None else self.reducer.after_launch, ss, hook_at=self.reducer.hook_indices())
        if ss is not None:
            ss.end()
        self.reducer.wait()
        mf = self.model
        on = t is not None and t.wants("optimizer")
        if on:
            t.begin("optimizer")
        _optimizer_step(self, mf, p)
        if on:
            t.end()

    def step(self, image, rand_indices=None):
        self.set_batch(image, rand_indices)
        self.run()
        return self.loss_out
