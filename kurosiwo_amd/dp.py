"""Data parallelism for the flat gradient arena: one process per GPU,
`torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The backward pass is a static launch list, so gradient readiness is known at plan-build time:
the arena is cut into contiguous buckets and each bucket's all-reduce(SUM) is issued right after
the last launch that writes into it.  ProcessGroupNCCL runs the collective on its own HIP stream
(it waits on the launching stream's event), i.e. the reduction overlaps the remaining backward
kernels; the optimiser waits for the handles and folds the 1/world average into its kernel.
"""
import os

import torch
import torch.distributed as dist

_FORCE = bool(os.environ.get("KSMI_DP_FORCE"))      # exercise the collective path in a 1-rank group (single-GPU smoke test of RCCL)


def make_buckets(ready_index, offsets, numels, total, bucket_elems):
    """ready_index[key] = index of the last backward launch writing parameter `key`.
    Returns [(start, end, after_launch)] covering [0, total) in arena order."""
    keys = sorted(offsets, key=lambda k: offsets[k])
    buckets, start, after = [], 0, -1
    for i, k in enumerate(keys):
        after = max(after, ready_index.get(k, -1))
        end = offsets[keys[i + 1]] if i + 1 < len(keys) else total
        if end - start >= bucket_elems or i + 1 == len(keys):
            buckets.append((start, end, after))
            start, after = end, -1
    return buckets


def default_grad_dtype(numel):
    """wire format of the gradient buckets.  Default fp32 (what the reference's optimiser would see from a DDP reduction); "bf16" is
    opt-in (configs["dp_grad_dtype"] / KSMI_DP_GRAD_DTYPE=bf16): it halves the bytes a ring moves over xGMI (7 links x ~153 GB/s per
    GPU, per-link bound; FloodViT 822 MB of fp32 gradients: ring all-reduce ~9.4 ms against a ~14 ms step) but sums in bf16 and
    overwrites the local fp32 gradients with the rounded sums, so it stays off until a world > 1 K-step gate has passed on hardware
    (tests/test_gpu_dp.py::test_two_gpu_*).  For large arenas prefer mode="rs_ag" (BucketedAllReduce), which keeps fp32."""
    env = os.environ.get("KSMI_DP_GRAD_DTYPE")
    if env in ("fp32", "bf16"):
        return env
    return "fp32"


def default_mode():
    """collective per bucket: "all_reduce" (default) or "rs_ag" = reduce-scatter + all-gather of the bucket in place (SURVEY.md §5 /
    §8(e): every rank reduces 1/W of the bucket, so on the fully connected xGMI node both phases can use all 7 links of a GPU instead
    of a ring's one; fp32 on the wire, same sums).  configs["dp_mode"] / KSMI_DP_MODE select it."""
    env = os.environ.get("KSMI_DP_MODE")
    return env if env in ("all_reduce", "rs_ag") else "all_reduce"


class BucketedAllReduce:
    """grad_dtype = "bf16": a bucket is cast into a bf16 staging buffer behind its last writer, the staging buffer is all-reduced (SUM
    in bf16 on the wire and in RCCL's reduction), and wait() casts the sums back into the fp32 arena the optimiser reads; the fp32
    master gradients of the local rank are lost to that rounding too (every rank ends with the same bits).

    mode = "rs_ag": reduce_scatter_tensor + all_gather_into_tensor on the bucket in place (the leading len - len % W elements; the
    < W tail elements ride on a small all-reduce); gloo has no reduce-scatter, there the mode degenerates to all-reduce.

    Streams: a bucket may hold gradients written on any stream of the step (compute lanes, weight-gradient side stream).  With
    `writer_streams` given (StepStreams.all_streams) the collective is issued from a dedicated issue stream that waits for an event
    recorded on each of them, so that no compute stream ever waits for another one because a bucket became ready; without it the
    collective is issued on the current stream (single-stream steps)."""

    def __init__(self, flat_grads, buckets, group=None, grad_dtype="fp32", mode=None):
        self.flat, self.group = flat_grads, group
        self.buckets = buckets
        self.grad_dtype = grad_dtype
        self.mode = mode or default_mode()
        if self.mode not in ("all_reduce", "rs_ag"):
            raise ValueError(f"dp mode {self.mode!r}: all_reduce | rs_ag")
        self.stage = torch.empty(flat_grads.numel(), dtype=torch.bfloat16, device=flat_grads.device) if grad_dtype == "bf16" else None
        self.staged = []
        self.by_launch = {}
        for b in buckets:
            self.by_launch.setdefault(b[2], []).append(b)
        self.handles = []
        self.issued = []
        self.issue_stream = None
        self._events = {}                  # writer stream handle -> its reusable event (_issue)
        self.writer_streams = None         # callable -> streams whose work a ready bucket may depend on

    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def active(self):
        return self.world() > 1 or (_FORCE and dist.is_initialized())

    def _collective(self, t):
        """SUM of `t` over the ranks, in place, asynchronously; returns the handles"""
        W = self.world()
        if self.mode == "rs_ag" and dist.get_backend(self.group) != "gloo":
            n = t.numel() - t.numel() % W
            hs = []
            if n:
                r = dist.get_rank(self.group)
                shard = t[r * (n // W):(r + 1) * (n // W)]
                hs.append(dist.reduce_scatter_tensor(shard, t[:n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                hs.append(dist.all_gather_into_tensor(t[:n], shard, group=self.group, async_op=True))
            if n < t.numel():
                hs.append(dist.all_reduce(t[n:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return hs
        return [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]

    def _issue(self, fn):
        """run fn() (which enqueues the casts / collectives of one bucket) on the issue stream behind every writer stream"""
        streams = self.writer_streams() if self.writer_streams is not None else None
        if not streams or not self.flat.is_cuda:
            return fn()
        if self.issue_stream is None:
            self.issue_stream = torch.cuda.Stream(device=self.flat.device)
        for s in streams:
            # one event per writer stream for the life of the reducer: hipStreamWaitEvent captures the record it finds when it is
            # enqueued, so the event may be re-recorded for the next bucket as soon as this call returns (no create / destroy per
            # bucket per stream per step)
            ev = self._events.get(s.cuda_stream)
            if ev is None:
                ev = self._events[s.cuda_stream] = torch.cuda.Event()
            ev.record(s)
            self.issue_stream.wait_event(ev)
        with torch.cuda.stream(self.issue_stream):
            return fn()

    def hook_indices(self):
        """indices of the backward launch list at which after_launch has a bucket to issue (none while no collective is active: the
        compiled launch list then runs in one piece, snunet_plan.LaunchList.run)"""
        return sorted(k for k in self.by_launch if k >= 0) if self.active() else []

    def after_launch(self, idx):
        for (s, e, _) in self.by_launch.get(idx, ()):
            self.issued.append((s, e))
            if self.active():
                view = self.flat[s:e]
                if dist.get_backend(self.group) == "gloo" and view.is_cuda:       # CPU-backend tests with device gradients
                    for st in (self.writer_streams() if self.writer_streams is not None else ()):
                        st.synchronize()
                    torch.cuda.current_stream().synchronize()
                    host = view.cpu()
                    if self.stage is not None:                                   # same wire rounding as the device path
                        host = host.to(torch.bfloat16)
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                    view.copy_(host)
                elif self.stage is not None:
                    st = self.stage[s:e]

                    def go(st=st, view=view):
                        st.copy_(view)                                           # fp32 -> bf16 behind the last writer
                        return self._collective(st)
                    self.handles += self._issue(go)
                    self.staged.append((s, e))
                else:
                    self.handles += self._issue(lambda view=view: self._collective(view))

    def flush(self):
        """issue every bucket that has not been issued yet (a bucket whose readiness index was never reached: a parameter
        without a registered gradient writer must still be averaged)"""
        done = set(self.issued)
        for (s, e, after) in self.buckets:
            if (s, e) not in done:
                self.by_launch.setdefault(-2, []).append((s, e, -2))
        if -2 in self.by_launch:
            self.after_launch(-2)
            del self.by_launch[-2]

    def wait(self):
        """the current stream continues behind every collective of the step"""
        self.flush()
        for h in self.handles:
            h.wait()
        if self.issue_stream is not None and self.handles:
            torch.cuda.current_stream().wait_stream(self.issue_stream)
        for (s, e) in self.staged:
            self.flat[s:e].copy_(self.stage[s:e])                                # bf16 sums -> the fp32 arena
        self.handles = []
        self.issued = []
        self.staged = []
