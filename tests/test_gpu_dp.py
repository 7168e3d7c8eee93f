"""Data parallelism on a real model: two ranks training on the two halves of a batch must take the
same optimiser step as one rank on the whole batch.  Both ranks share the one GPU of the test box, so the process group is gloo
(device gradients staged through the host by kurosiwo_amd/dp.py); on a multi-GPU node the identical code runs over RCCL.
BN-free model (FloodViT, fp32): equality to rounding.  SNUNet keeps per-rank BatchNorm statistics (the reference has no SyncBN),
so its two-rank step equals the single-rank step only through the BN-free parameters; that is checked as a documented property."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SMALL = dict(channels=6, image_size=224, patch_size=16, dim=1024, depth=2, heads=4, mlp_dim=512)
CFG = {"mlp": False, "decoder": True, "num_classes": 3, "image_size": 224, "finetuning_patch_size": 16}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    from kurosiwo_amd.floodvit import FinetunerSegmentation, ViT
    from oracle import vit_ref as V
    from oracle.seeded import seeded_fill_
    hp = SMALL
    enc = ViT(image_size=hp["image_size"], patch_size=hp["patch_size"], num_classes=1000, dim=hp["dim"], depth=hp["depth"],
              heads=hp["heads"], mlp_dim=hp["mlp_dim"], channels=hp["channels"])
    model = FinetunerSegmentation(enc, CFG, precision="fp32")
    model.load_state_dict(seeded_fill_(V.new_state_dict(**hp)))
    return model.cuda().train()


def _data(B):
    from oracle.seeded import seeded_labels, seeded_tensor
    x = seeded_tensor("dp.x", (B, 6, 224, 224)).clamp_(-2.23, 5.75)
    lbl = seeded_labels("dp.lbl", (B, 224, 224), p_invalid=0.0)        # no ignored pixels: every shard normalises by the same count
    return x, lbl


def _steps(model, x, lbl, n):
    from kurosiwo_amd.optim import FusedSGD
    from kurosiwo_amd.trainer import SegTrainStep
    # plain SGD: the parameter update is linear in the gradient (Adam turns rounding noise of near-zero gradients into +-lr steps)
    step = SegTrainStep(model, x.shape[0], "cross_entropy", [1.0, 1.0, 1.0], optimizer=FusedSGD(model.parameters(), lr=0.05),
                        bucket_mb=16.0)
    losses = []
    for _ in range(n):
        losses.append(float(step.step(x.cuda(), lbl.cuda())[0]))
    torch.cuda.synchronize()
    return losses, step


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kurosiwo_amd import distributed as D
    model = _build()
    if rank == 1:
        model.flat_params.mul_(1.5)                     # a rank that started from different weights ...
    D.broadcast_model_(model)                           # ... is overwritten by rank 0's
    x, lbl = _data(4)
    xs, ls = D.shard_batch((x, lbl), rank, world)
    losses, step = _steps(model, xs, ls, 2)
    covered = sum(e - s for s, e, _ in step.reducer.buckets)
    q.put((rank, losses, model.flat_params.detach().cpu().numpy(), covered == model.flat_params.numel()))   # (by value)
    dist.destroy_process_group()


def test_two_rank_step_equals_single_rank_step_on_the_whole_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model = _build()
    p0 = model.flat_params.detach().cpu().clone()
    x, lbl = _data(4)
    losses, _ = _steps(model, x, lbl, 2)
    ref = model.flat_params.detach().cpu()
    (r0, l0, w0, c0), (r1, l1, w1, c1) = res
    w0, w1 = torch.from_numpy(w0), torch.from_numpy(w1)
    assert c0 and c1, "gradient buckets must cover the whole arena"
    assert torch.equal(w0, w1), "ranks diverged"
    upd = (ref - p0).abs().max()
    assert float((w0 - ref).abs().max()) < 1e-3 * float(upd) + 1e-7, (float((w0 - ref).abs().max()), float(upd))
    # mean of the shard losses = loss of the whole batch (equal shard sizes, no ignored pixels)
    for k in range(2):
        assert abs(0.5 * (l0[k] + l1[k]) - losses[k]) < 2e-4 * abs(losses[k]), (k, l0[k], l1[k], losses[k])


def _snunet_build(sync_bn):
    from kurosiwo_amd.snunet import SNUNet_ECAM
    from oracle import snunet_ref as R
    from oracle.seeded import seeded_fill_
    m = SNUNet_ECAM(2, 3, base_channel=16, precision="fp32")
    m.sync_bn = sync_bn
    m.load_state_dict(seeded_fill_(R.new_state_dict(2, 3, 16)))
    return m.cuda().train()


def _snunet_data(B):
    from oracle.seeded import seeded_labels, seeded_tensor
    xA, xB = seeded_tensor("syncbn.xA", (B, 2, 64, 64)), seeded_tensor("syncbn.xB", (B, 2, 64, 64))
    lbl = seeded_labels("syncbn.lbl", (B, 64, 64), p_invalid=0.0)       # no ignored pixels: every shard normalises by the same count
    return xA, xB, lbl


def _snunet_steps(model, xA, xB, lbl, n):
    from kurosiwo_amd.optim import FusedSGD
    from kurosiwo_amd.trainer import CDTrainStep
    step = CDTrainStep(model, xA.shape[0], 64, 64, loss_function="ce+dice", class_weights=(1.0, 1.0, 1.0),
                       optimizer=FusedSGD(model.parameters(), lr=0.05))
    losses = [float(step.step(xA.cuda(), xB.cuda(), lbl.cuda())[0]) for _ in range(n)]
    torch.cuda.synchronize()
    return losses


def _syncbn_worker(rank, world, port, q, sync_bn):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kurosiwo_amd import distributed as D
    model = _snunet_build(sync_bn)
    D.broadcast_model_(model)
    xA, xB, lbl = _snunet_data(4)
    a, b, l = D.shard_batch((xA, xB, lbl), rank, world)
    losses = _snunet_steps(model, a, b, l, 2)
    q.put((rank, losses, model.flat_params.detach().cpu().numpy(), model.flat_buffers.detach().cpu().numpy()))
    dist.destroy_process_group()


def test_syncbn_two_ranks_equal_single_process():
    """SURVEY.md §8(e) "second-order items": with the optional SyncBN switch (SNUNet_ECAM.sync_bn / KSMI_SYNC_BN=1 / configs["sync_bn"]:
    every BatchNorm call all-reduces its statistics rows, forward (sum, sum of squares) and backward (sum g, sum g xhat), and finishes
    them with the global pixel count) two ranks on the two halves of a batch take the optimiser steps ONE process takes on the whole
    batch: parameters AND BatchNorm running statistics agree to fp32 rounding after two steps.  Without the switch (the reference's
    per-process BatchNorm, models/snunet.py:16,18) the same run ends somewhere else -- the control that the switch is what is tested.
    Both ranks share the test box's one GPU, so the group is gloo; the collective calls are the ones RCCL serves on a node."""
    ctx = mp.get_context("spawn")
    out = {}
    for sync_bn in (True, False):
        q, port = ctx.Queue(), _free_port()
        procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q, sync_bn)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        out[sync_bn] = res
    model = _snunet_build(False)
    p0 = model.flat_params.detach().cpu().clone()
    xA, xB, lbl = _snunet_data(4)
    losses = _snunet_steps(model, xA, xB, lbl, 2)
    ref_p, ref_b = model.flat_params.detach().cpu(), model.flat_buffers.detach().cpu()
    upd = float((ref_p - p0).abs().max())
    (r0, l0, w0, b0), (r1, l1, w1, b1) = out[True]
    w0, w1, b0, b1 = map(torch.from_numpy, (w0, w1, b0, b1))
    assert torch.equal(w0, w1) and torch.equal(b0, b1), "ranks diverged under SyncBN"
    dp, db = float((w0 - ref_p).abs().max()), float((b0 - ref_b).abs().max())
    print(f"SyncBN two ranks vs one process: max |d param| {dp:.3e} (largest update {upd:.3e}), max |d running stat| {db:.3e}; "
          f"losses {l0} {l1} vs {losses}")
    assert dp < 2e-3 * upd + 1e-6, (dp, upd)
    assert db < 1e-5 * float(ref_b.abs().max()) + 1e-6, db
    for k in range(2):                                  # mean of the shard losses = loss of the whole batch (same statistics on both sides)
        assert abs(0.5 * (l0[k] + l1[k]) - losses[k]) < 2e-4 * abs(losses[k]), (k, l0[k], l1[k], losses[k])
    # control: per-rank BatchNorm (the default) is a different function of the same data
    (_, _, v0, _), _ = out[False]
    d_plain = float((torch.from_numpy(v0) - ref_p).abs().max())
    print(f"per-rank BatchNorm (default) two ranks vs one process: max |d param| {d_plain:.3e}")
    assert d_plain > 20 * dp, (d_plain, dp)


def test_every_parameter_with_a_gradient_has_a_readiness_index():
    """A parameter missing from plan.param_ready would be reduced before (or never after) its writer ran."""
    from kurosiwo_amd.snunet import SNUNet_ECAM
    m = SNUNet_ECAM(2, 3, base_channel=16, precision="bf16").cuda().train()
    plan = m.plan(2, 32, 32, True, True)
    missing = [k for k in m._poff if k not in plan.param_ready]
    assert not missing, missing
    assert max(plan.param_ready.values()) == len(plan.bwd.calls) - 1


def test_main_entry_under_torchrun_two_ranks(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 main.py --method snunet ...`: both ranks join the group, shard every batch,
    all-reduce gradients / confusion matrices, rank 0 alone writes the checkpoints, every rank reports the same test mIoU."""
    import re
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copytree(os.path.join(root, "configs"), tmp_path / "configs")
    env = dict(os.environ, KSMI_SYNTHETIC_TILES="8,4,4", KSMI_DIST_BACKEND="gloo", PYTHONPATH=root, MASTER_ADDR="127.0.0.1")
    wrapper = tmp_path / "run_main.py"
    wrapper.write_text("import os, sys\nsys.path.insert(0, %r)\nimport main\nm = main.main(sys.argv[1:])\n"
                       # one write() per rank: print() issues one per argument and the two ranks share the pipe
                       "sys.stdout.write('\\nRANK %%s MIOU %%r\\n' %% (os.environ['RANK'], float(m))); sys.stdout.flush()\n" % root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(wrapper), "--method", "snunet", "--inputs", "pre_event_1", "post_event", "--batch_size", "4"]
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    mious = dict(re.findall(r"RANK (\d) MIOU ([0-9.e+-]+)", out.stdout))
    assert set(mious) == {"0", "1"}, out.stdout[-2000:]
    assert mious["0"] == mious["1"], mious
    runs = list((tmp_path / "checkpoints" / "snunet").glob("*"))
    assert len(runs) == 1, runs                      # one time-stamped directory, named by rank 0
    assert (runs[0] / "best_segmentation.pt").exists()
    assert out.stdout.count("Samples in Train Set") >= 1


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_main_entry_on_the_rccl_backend_one_rank(tmp_path, wire):
    """The REAL backend on the one GPU of the test box: `KSMI_DP_FORCE=1 python main.py --method snunet` joins a one-rank "nccl" (= RCCL)
    group (init_process_group with device_id), and every gradient bucket goes through the bucket hooks, the join of the step's three
    streams and an RCCL all-reduce issued during backward.  A SUM over one rank is the identity: with fp32 buckets the run must end at
    exactly the mIoU of the run without collectives; with bf16 buckets (the wire format FloodViT defaults to) the gradients are rounded
    to bf16 once, so the run only has to stay healthy."""
    import re
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wrapper = tmp_path / "run_main.py"
    wrapper.write_text("import sys\nsys.path.insert(0, %r)\nimport main\nm = main.main(sys.argv[1:])\n"
                       "import torch.distributed as dist\n"
                       "print('BACKEND', dist.get_backend() if dist.is_initialized() else 'none', 'MIOU', repr(float(m)), flush=True)\n" % root)
    res = {}
    for tag, extra in (("plain", {}), ("rccl", {"KSMI_DP_FORCE": "1", "KSMI_DP_GRAD_DTYPE": wire, "MASTER_PORT": str(_free_port())})):
        wd = tmp_path / tag
        wd.mkdir()
        shutil.copytree(os.path.join(root, "configs"), wd / "configs")
        env = dict(os.environ, KSMI_SYNTHETIC_TILES="8,4,4", PYTHONPATH=root, MASTER_ADDR="127.0.0.1", **extra)
        env.pop("KSMI_DIST_BACKEND", None)
        out = subprocess.run([sys.executable, str(wrapper), "--method", "snunet", "--inputs", "pre_event_1", "post_event", "--batch_size", "4"],
                             cwd=wd, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        res[tag] = re.findall(r"BACKEND (\w+) MIOU ([0-9.e+-]+)", out.stdout)[-1]
    assert res["plain"][0] == "none" and res["rccl"][0] == "nccl", res
    if wire == "fp32":
        assert res["plain"][1] == res["rccl"][1], res
    else:
        assert 0.0 <= float(res["rccl"][1]) <= 100.0 and abs(float(res["rccl"][1]) - float(res["plain"][1])) < 15.0, res


def test_rs_ag_bucket_mode_on_the_rccl_backend_one_rank(tmp_path):
    """mode = "rs_ag" (reduce_scatter_tensor + all_gather_into_tensor in place, kurosiwo_amd/dp.py) through the REAL backend on the one
    GPU of the test box: over one rank both collectives are identities, so the run must end at exactly the mIoU of the plain run."""
    import re
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wrapper = tmp_path / "run_main.py"
    wrapper.write_text("import sys\nsys.path.insert(0, %r)\nimport main\nm = main.main(sys.argv[1:])\n"
                       "print('MIOU', repr(float(m)), flush=True)\n" % root)
    res = {}
    for tag, extra in (("plain", {}), ("rs_ag", {"KSMI_DP_FORCE": "1", "KSMI_DP_MODE": "rs_ag", "MASTER_PORT": str(_free_port())})):
        wd = tmp_path / tag
        wd.mkdir()
        shutil.copytree(os.path.join(root, "configs"), wd / "configs")
        env = dict(os.environ, KSMI_SYNTHETIC_TILES="8,4,4", PYTHONPATH=root, MASTER_ADDR="127.0.0.1", **extra)
        env.pop("KSMI_DIST_BACKEND", None)
        out = subprocess.run([sys.executable, str(wrapper), "--method", "snunet", "--inputs", "pre_event_1", "post_event", "--batch_size", "4"],
                             cwd=wd, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        res[tag] = re.findall(r"MIOU ([0-9.e+-]+)", out.stdout)[-1]
    assert res["plain"] == res["rs_ag"], res


# ---- two or more GPUs: these tests activate themselves the moment a multi-GPU node runs the suite (the single-GPU test box skips them) ----
needs2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL over xGMI); the 1-GPU box covers the same path over gloo / one-rank RCCL")


def _torchrun_bench(nproc, extra_env=None, args=()):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--no-solo", *args]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                    # ONE JSON line, from rank 0
    return json.loads(lines[0])


@needs2
@pytest.mark.parametrize("model,extra", [("snunet", ()), ("floodvit", ()), ("changeformer", ("--channels", "4"))])
def test_two_gpus_bench_on_rccl(model, extra):
    """`torchrun --nproc-per-node 2 bench.py --gpus 2` on the nccl (= RCCL) backend: two ranks, different tiles per rank, and after the
    timed steps the parameter arenas of the ranks agree bit for bit (bench.py's dp_check: the bucketed all-reduce really averaged the
    gradients), every value finite, whole-job throughput reported for 2 GPUs."""
    r = _torchrun_bench(2, args=("--model", model, *extra))
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "dp2"
    chk = r["config"]["dp_check"]
    assert chk["ranks"] == 2 and chk["backend"] == "nccl"
    assert chk["params_equal"] and chk["params_finite"], chk
    assert chk["rank_losses"][0] != chk["rank_losses"][1], "the ranks must train on different tiles"
    print(model, "2 x MI355X:", r["value"], r["unit"], r["ms_per_step"], "ms/step", chk)


@needs2
def test_two_gpus_rccl_equals_the_gloo_path():
    """the same two-rank run with the collectives on gloo (host staging): a SUM of two fp32 values does not depend on the order, so
    RCCL and gloo must leave the SAME parameter bits (integer checksum) and per-rank losses"""
    a = _torchrun_bench(2)
    b = _torchrun_bench(2, {"KSMI_DIST_BACKEND": "gloo"})
    ca, cb = a["config"]["dp_check"], b["config"]["dp_check"]
    assert ca["backend"] == "nccl" and cb["backend"] == "gloo"
    assert ca["params_equal"] and cb["params_equal"]
    assert ca["param_checksum"] == cb["param_checksum"], (ca, cb)
    assert ca["rank_losses"] == cb["rank_losses"], (ca, cb)


@needs2
@pytest.mark.parametrize("mode,wire", [("rs_ag", "fp32"), ("all_reduce", "bf16")])
def test_two_gpus_other_bucket_modes(mode, wire):
    """reduce-scatter + all-gather buckets give the all-reduce parameters (fp32 sums over two ranks: bit-identical); the opt-in bf16 wire
    format keeps the ranks identical to each other and the run healthy (the K-step gate for making it a default is a scaling-node job)"""
    ref = _torchrun_bench(2)
    r = _torchrun_bench(2, {"KSMI_DP_MODE": mode, "KSMI_DP_GRAD_DTYPE": wire})
    chk = r["config"]["dp_check"]
    assert chk["params_equal"] and chk["params_finite"] and chk["dp_mode"] == mode and chk["grad_wire"] == wire, chk
    if wire == "fp32":
        assert chk["param_checksum"] == ref["config"]["dp_check"]["param_checksum"], (chk, ref["config"]["dp_check"])
    else:
        for x, y in zip(chk["rank_losses"], ref["config"]["dp_check"]["rank_losses"]):
            assert abs(x - y) < 0.05 * abs(y) + 1e-3, (chk, ref["config"]["dp_check"])


# ---- the FULL fused train step under data parallelism for the other two bucket plans (FloodViT: 280 keys,
# ChangeFormer: 373 keys, BatchNorm in the decoder), with the opt-in wire settings -- mode "rs_ag" (gloo has no reduce-scatter: the mode
# degrades to all-reduce, the bucket bookkeeping is the same) and the bf16 wire -- two ranks on the halves of a batch, both on the test
# box's one GPU over gloo, three optimiser steps, multi-stream step (side-stream weight gradients) and the compiled launch list.
def _family_build(family, precision):
    torch.manual_seed(11)
    if family == "floodvit":
        from kurosiwo_amd.floodvit import FinetunerSegmentation, ViT
        hp = SMALL
        enc = ViT(image_size=hp["image_size"], patch_size=hp["patch_size"], num_classes=1000, dim=hp["dim"], depth=hp["depth"],
                  heads=hp["heads"], mlp_dim=hp["mlp_dim"], channels=hp["channels"])
        return FinetunerSegmentation(enc, CFG, precision=precision).cuda().train()
    from kurosiwo_amd.changeformer import ChangeFormerV6
    m = ChangeFormerV6(input_nc=2, output_nc=3, decoder_softmax=True, embed_dim=64, precision=precision)
    for k in ("drop_rate", "attn_drop", "drop_path_rate"):             # (regulariser-free: the ranks would draw from one counter-based stream anyway)
        if hasattr(m, k):
            setattr(m, k, 0.0)
    return m.cuda().train()


def _family_steps(family, model, B, rank, world, wire, mode, n):
    from kurosiwo_amd import distributed as D
    from kurosiwo_amd.optim import FusedSGD
    from kurosiwo_amd.trainer import CDTrainStep, SegTrainStep
    from oracle.seeded import seeded_labels, seeded_tensor
    S = 224                                                              # (ChangeFormer's SR attention is specialised for 224 x 224 tiles)
    lbl = seeded_labels("dp2.lbl", (B, S, S), p_invalid=0.0)
    opt = FusedSGD(model.parameters(), lr=0.02)
    kw = dict(optimizer=opt, bucket_mb=2.0, grad_dtype=wire, dp_mode=mode)
    if family == "floodvit":
        x = seeded_tensor("dp2.x", (B, 6, S, S)).clamp_(-2.23, 5.75)
        (xs, ls) = D.shard_batch((x, lbl), rank, world) if world > 1 else (x, lbl)
        step = SegTrainStep(model, xs.shape[0], "cross_entropy", [1.0, 1.0, 1.0], **kw)
        args = (xs.cuda(), ls.cuda())
    else:
        xA, xB = seeded_tensor("dp2.xA", (B, 2, S, S)), seeded_tensor("dp2.xB", (B, 2, S, S))
        (a, b, ls) = D.shard_batch((xA, xB, lbl), rank, world) if world > 1 else (xA, xB, lbl)
        step = CDTrainStep(model, a.shape[0], S, S, "ce+dice", (1.0, 1.0, 1.0), **kw)
        args = (a.cuda(), b.cuda(), ls.cuda())
    losses = [float(step.step(*args)[0]) for _ in range(n)]
    torch.cuda.synchronize()
    return losses, step


def _family_worker(rank, world, port, q, family, precision, wire, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kurosiwo_amd import distributed as D
    model = _family_build(family, precision)
    if rank == 1:
        model.flat_params.mul_(1.25)
    D.broadcast_model_(model)
    losses, step = _family_steps(family, model, 4, rank, world, wire, mode, 3)
    missing = [k for k in model._poff if k not in step.plan.param_ready]
    covered = sum(e - s for s, e, _ in step.reducer.buckets) == model.flat_params.numel()
    q.put((rank, losses, model.flat_params.detach().cpu().numpy(), covered, missing, len(step.reducer.buckets), step.reducer.hook_indices(),
           step._ss is not None and step._ss._runner is not None))
    dist.destroy_process_group()


@pytest.mark.parametrize("family,precision,wire,mode", [("floodvit", "fp32", "fp32", "rs_ag"), ("floodvit", "bf16", "bf16", "all_reduce"),
                                                        ("changeformer", "fp32", "fp32", "rs_ag"), ("changeformer", "bf16", "bf16", "all_reduce")])
def test_two_ranks_full_train_step_other_bucket_plans(family, precision, wire, mode):
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_family_worker, args=(r, 2, port, q, family, precision, wire, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    import queue as _queue
    import time as _time
    t_end = _time.time() + 600
    while len(res) < 2:                                                  # (a worker that died must fail the test at once, not after the queue time-out)
        try:
            res.append(q.get(timeout=2))
        except _queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or _time.time() > t_end:
                for p in procs:
                    p.kill()
                pytest.fail(f"worker exit codes {dead or 'time-out'}")
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, l0, w0, c0, m0, nb0, hooks0, fast0), (r1, l1, w1, c1, m1, nb1, hooks1, fast1) = res
    w0, w1 = torch.from_numpy(w0), torch.from_numpy(w1)
    assert c0 and c1 and not m0 and not m1, (c0, c1, m0[:5], m1[:5])          # buckets cover the arena; every parameter has a readiness index
    assert nb0 == nb1 >= 2 and hooks0 == hooks1 and len(hooks0) >= 2            # several buckets issued DURING the backward list ...
    assert fast0 and fast1                                                      # ... from the compiled launch list, cut at the bucket indices
    assert torch.equal(w0, w1), f"ranks diverged: max |d| {float((w0 - w1).abs().max()):.3e}"
    assert all(abs(x) < 1e6 for x in l0 + l1) and torch.isfinite(w0).all()
    if precision == "fp32" and family == "floodvit":
        # BN-free model, fp32 sums on the wire: the two ranks took the step ONE process takes on the whole batch
        model = _family_build(family, precision)
        p0 = model.flat_params.detach().cpu().clone()
        losses, _ = _family_steps(family, model, 4, 0, 1, "fp32", None, 3)
        ref = model.flat_params.detach().cpu()
        upd = float((ref - p0).abs().max())
        assert float((w0 - ref).abs().max()) < 2e-3 * upd + 1e-7, (float((w0 - ref).abs().max()), upd)
        for k in range(3):
            assert abs(0.5 * (l0[k] + l1[k]) - losses[k]) < 3e-4 * abs(losses[k]), (k, l0[k], l1[k], losses[k])
    else:
        # per-rank BatchNorm (ChangeFormer's decoder) and / or bf16 sums: a different function of the same data by design; the losses of the
        # shards still descend together
        assert 0.5 * (l0[-1] + l1[-1]) < 0.5 * (l0[0] + l1[0]) * 1.05, (l0, l1)
