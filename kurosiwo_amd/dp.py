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


class BucketedAllReduce:
    """grad_dtype = "bf16": a bucket is cast into a bf16 staging buffer behind its last writer, the staging buffer is all-reduced (SUM
    in bf16 on the wire and in RCCL's reduction), and wait() casts the sums back into the fp32 arena the optimiser reads; the fp32
    master gradients of the local rank are lost to that rounding too (every rank ends with the same bits)."""

    def __init__(self, flat_grads, buckets, group=None, grad_dtype="fp32"):
        self.flat, self.group = flat_grads, group
        self.buckets = buckets
        self.grad_dtype = grad_dtype
        self.stage = torch.empty(flat_grads.numel(), dtype=torch.bfloat16, device=flat_grads.device) if grad_dtype == "bf16" else None
        self.staged = []
        self.by_launch = {}
        for b in buckets:
            self.by_launch.setdefault(b[2], []).append(b)
        self.handles = []
        self.issued = []

    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def after_launch(self, idx):
        for (s, e, _) in self.by_launch.get(idx, ()):
            self.issued.append((s, e))
            if self.world() > 1 or (_FORCE and dist.is_initialized()):
                view = self.flat[s:e]
                if dist.get_backend(self.group) == "gloo" and view.is_cuda:       # CPU-backend tests with device gradients
                    torch.cuda.current_stream().synchronize()
                    host = view.cpu()
                    if self.stage is not None:                                   # same wire rounding as the device path
                        host = host.to(torch.bfloat16)
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                    view.copy_(host)
                elif self.stage is not None:
                    st = self.stage[s:e]
                    st.copy_(view)                                               # fp32 -> bf16 on the issuing stream, behind the last writer
                    self.handles.append(dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    self.staged.append((s, e))
                else:
                    self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

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
        self.flush()
        for h in self.handles:
            h.wait()
        for (s, e) in self.staged:
            self.flat[s:e].copy_(self.stage[s:e])                                # bf16 sums -> the fp32 arena
        self.handles = []
        self.issued = []
        self.staged = []
