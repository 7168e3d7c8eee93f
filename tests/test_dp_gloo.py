"""CPU, world_size 2, gloo: the data-parallel gradient path of kurosiwo_amd/dp.py (bucket
construction from launch-readiness indices, overlapped all-reduce issue order, SUM semantics).
The same code runs over RCCL ("nccl") on the GPUs; only the backend differs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kurosiwo_amd.dp import BucketedAllReduce, make_buckets


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_make_buckets_cover_arena_in_order():
    offsets = {"a": 0, "b": 100, "c": 300, "d": 1000}
    ready = {"a": 9, "b": 7, "c": 3, "d": 1}
    b = make_buckets(ready, offsets, None, 1200, 250)
    assert b[0][0] == 0 and b[-1][1] == 1200
    for (s0, e0, _), (s1, e1, _) in zip(b, b[1:]):
        assert e0 == s1
    assert b == [(0, 300, 9), (300, 1000, 3), (1000, 1200, 1)]
    assert make_buckets(ready, offsets, None, 1200, 10 ** 9) == [(0, 1200, 9)]


def _worker(rank, world, port, q, grad_dtype="fp32", mode=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1200
    flat = torch.zeros(n)
    offsets = {"a": 0, "b": 100, "c": 300, "d": 1000}
    ready = {"a": 9, "b": 7, "c": 3, "d": 1}          # backward finishes the arena tail first
    red = BucketedAllReduce(flat, make_buckets(ready, offsets, None, n, 250), grad_dtype=grad_dtype, mode=mode)
    order = []
    for launch in range(10):                          # the "backward launch list"
        if launch == 1:
            flat[1000:1200] = rank + 1.0
        if launch == 3:
            flat[300:1000] = 10.0 * (rank + 1)
        if launch == 7:
            flat[100:300] = 100.0 * (rank + 1)
        if launch == 9:
            flat[0:100] = -1.0 * (rank + 1)
        before = len(red.issued)
        red.after_launch(launch)
        order += [(launch, s, e) for (s, e) in red.issued[before:]]
    red.wait()
    tot = sum(r + 1.0 for r in range(world))
    ok = (torch.allclose(flat[1000:], torch.full((200,), tot)) and torch.allclose(flat[300:1000], torch.full((700,), 10 * tot))
          and torch.allclose(flat[100:300], torch.full((200,), 100 * tot)) and torch.allclose(flat[:100], torch.full((100,), -tot)))
    q.put((rank, ok, order))
    dist.destroy_process_group()


@pytest.mark.parametrize("grad_dtype,mode", [("fp32", None), ("bf16", None), ("fp32", "rs_ag")])
def test_bucketed_allreduce_world2_gloo(grad_dtype, mode):
    """grad_dtype = "bf16": the buckets travel as bf16 (staging buffer, SUM on the wire format, cast back into the fp32 arena in wait());
    the test values are exactly representable, so both wire formats give the same sums.  mode = "rs_ag": gloo has no reduce-scatter, the
    mode falls back to the all-reduce there (the RCCL form runs in tests/test_gpu_dp.py)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, grad_dtype, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, order in res:
        assert ok, rank
        assert order == [(1, 1000, 1200), (3, 300, 1000), (9, 0, 300)], order


def test_default_wire_format_is_fp32_and_bf16_is_opt_in(monkeypatch):
    """the reference's optimiser sees fp32 sums; the bf16 wire format changes optimiser numerics and is opt-in only"""
    from kurosiwo_amd.dp import default_grad_dtype
    monkeypatch.delenv("KSMI_DP_GRAD_DTYPE", raising=False)
    for n in (12_034_819, 41_035_255, 205_600_000):                 # SNUNet, ChangeFormer, FloodViT
        assert default_grad_dtype(n) == "fp32"
    monkeypatch.setenv("KSMI_DP_GRAD_DTYPE", "bf16")
    assert default_grad_dtype(205_600_000) == "bf16"


def test_single_process_is_a_noop():
    flat = torch.arange(10.0)
    red = BucketedAllReduce(flat, [(0, 10, 0)])
    red.after_launch(0)
    red.wait()
    assert torch.equal(flat, torch.arange(10.0))


def _syncbn_worker(rank, world, port, q):
    """the arithmetic of snunet_plan.SNUNetPlan.sync_bn on plain tensors: statistics rows summed over the ranks, finished with the
    GLOBAL count; backward sums (sum g, sum g xhat) all-reduced for d x while the BatchNorm parameter gradients stay local sums"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kurosiwo_amd import distributed as D
    g = torch.Generator().manual_seed(7)
    Bt, Cc, H = 4, 6, 5
    x = torch.randn(Bt, Cc, H, H, generator=g) * 2 + 1
    w, b = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    gy = torch.randn(Bt, Cc, H, H, generator=g)
    # whole batch, autograd (what one process computes)
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = torch.nn.functional.batch_norm(xr, None, None, wr, br, True, 0.1, 1e-5)
    (y * gy).sum().backward()
    # this rank's half, statistics through the collective
    sl = slice(rank * Bt // world, (rank + 1) * Bt // world)
    xs, gs = x[sl], gy[sl]
    n_local, n_glob = xs.numel() // Cc, x.numel() // Cc
    rows = torch.stack([xs.sum((0, 2, 3)), (xs * xs).sum((0, 2, 3))])          # one statistics row of this rank: [2][C]
    D.all_reduce_sum_(rows)
    mean = rows[0] / n_glob
    var = rows[1] / n_glob - mean * mean
    rstd = (var + 1e-5).rsqrt()
    xh = (xs - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    ys = xh * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    sums = torch.stack([gs.sum((0, 2, 3)), (gs * xh).sum((0, 2, 3))])          # local: these ARE the bias / weight gradients of this rank
    dw_local, db_local = sums[1].clone(), sums[0].clone()
    D.all_reduce_sum_(sums)                                                     # global: what d x needs
    dx = (w * rstd).view(1, -1, 1, 1) * (gs - (sums[0] / n_glob).view(1, -1, 1, 1) - xh * (sums[1] / n_glob).view(1, -1, 1, 1))
    dwg, dbg = dw_local.clone(), db_local.clone()
    D.all_reduce_sum_(dwg, dbg)                                                 # the gradient all-reduce of the DP step (SUM)
    ok = (torch.allclose(ys, y[sl].detach(), atol=1e-5) and torch.allclose(dx, xr.grad[sl], atol=1e-5)
          and torch.allclose(dwg, wr.grad, atol=1e-4) and torch.allclose(dbg, br.grad, atol=1e-4) and n_local * world == n_glob)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_syncbn_statistics_arithmetic_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
