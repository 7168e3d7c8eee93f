"""GPU parity of the whole SNUNet-ECAM train step (HIP kernels through the C-ABI) against the
CPU oracle, plus the committed golden vectors of the real reference at full size."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import snunet_ref as R
from oracle.seeded import seeded_fill_, seeded_labels, seeded_tensor

CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]


def sar_like(name, shape):
    return seeded_tensor(name, shape).clamp_(-2.23, 5.75)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(c, bc, precision, sd, dev):
    from kurosiwo_amd.snunet import SNUNet_ECAM
    m = SNUNet_ECAM(c, 3, base_channel=bc, precision=precision)
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    return m.to(dev)


@pytest.mark.parametrize("c,bc,B,H,W", [(2, 16, 2, 32, 32), (3, 16, 1, 48, 32), (2, 32, 2, 64, 64)])
def test_fp32_eval_logits(dev, c, bc, B, H, W):
    tag = f"ev{c}{bc}{B}{H}{W}"
    xA, xB = sar_like(tag + "A", (B, c, H, W)), sar_like(tag + "B", (B, c, H, W))
    sd = seeded_fill_(R.new_state_dict(c, 3, bc))
    with torch.no_grad():
        ref = R.snunet_forward(sd, xA, xB, training=False)
    m = _model(c, bc, "fp32", sd, dev).eval()
    with torch.no_grad():
        out = m(xA.to(dev), xB.to(dev)).cpu()
    rel = float((out - ref).abs().max() / ref.abs().max())
    assert rel < 1e-3, rel           # north-star: 1e-3 rel on logits (measured ~1e-6)
    assert rel < 5e-5, rel


@pytest.mark.parametrize("c,bc,B,H,W", [(2, 16, 2, 32, 32), (3, 16, 2, 32, 48), (3, 32, 1, 224, 224)])   # the last: configs[2] (VV, VH, DEM) at tile size
def test_fp32_train_step_matches_oracle(dev, c, bc, B, H, W):
    from kurosiwo_amd.loss import BCEandDiceLoss
    from kurosiwo_amd.optim import FusedAdam
    tag = f"tr{c}{bc}{B}{H}{W}"
    xA, xB = sar_like(tag + "A", (B, c, H, W)), sar_like(tag + "B", (B, c, H, W))
    lbl = seeded_labels(tag + "L", (B, H, W))
    sd = seeded_fill_(R.new_state_dict(c, 3, bc))
    m = _model(c, bc, "fp32", sd, dev).train()
    crit = BCEandDiceLoss(CLASS_WEIGHTS, 3, True)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    ref_opt = R.AdamRef(sd, lr=1e-3)
    for step in range(3):
        opt.zero_grad()
        logits = m(xA.to(dev), xB.to(dev))
        loss = crit(logits, lbl.to(dev))
        loss.backward()
        grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        opt.step()
        ref_loss, ref_logits, ref_grads = R.train_step(sd, ref_opt, xA, xB, lbl, CLASS_WEIGHTS, True)
        tol = 1e-4 * (10 ** step)        # trajectories drift apart slowly
        assert float((logits.detach().cpu() - ref_logits).abs().max() / ref_logits.abs().max()) < tol
        assert abs(float(loss) - ref_loss) < tol * max(1.0, abs(ref_loss))
        if step == 0:
            worst = 0.0
            for k, g in ref_grads.items():
                denom = float(g.abs().max())
                err = float((grads[k] - g).abs().max())
                if k.endswith("conv2.bias") or denom <= 1e-6:
                    # conv2 is followed by train-mode BatchNorm: its bias gradient is analytically 0 and
                    # both sides only hold rounding noise (reference ~1e-6, HIP ~1e-8)
                    assert err < 2e-5, (k, err)
                    continue
                worst = max(worst, err / denom)
                # a single ReLU-mask flip at |pre-activation| ~ 1e-7 moves a per-channel sum over only
                # B*H*W = 2k pixels by ~0.5 %: vectors get 1e-2, weight tensors 2e-3
                rtol = 2e-3 if g.dim() == 4 else 1e-2
                assert err < rtol * denom + 1e-6, (k, err, denom)
            print("worst relative grad error", worst)
            msd = m.state_dict()
            for k in sd:
                if k.endswith(("running_mean", "running_var")):
                    assert (msd[k].cpu() - sd[k]).abs().max() < 1e-4, k
                if k.endswith("num_batches_tracked"):
                    assert int(msd[k]) == int(sd[k]), k
    msd = m.state_dict()
    for k in R.param_keys(sd):
        if k.endswith("conv2.bias"):
            continue     # zero-gradient parameter: Adam turns rounding noise into +-lr steps on BOTH sides
        # Adam normalises every element's update to ~lr regardless of |g|: an element whose gradient is
        # rounding noise (dead ReLU channel, ...) may step +-lr in opposite directions on the two sides.
        # So: hard bound 2*lr*steps on every element, and >= 95 % of the elements within 2e-4 (the optimiser
        # kernel itself is checked bit-tightly in test_gpu_kernels.py::test_adam_and_sgd_match_reference_formulas).
        diff = (msd[k].cpu() - sd[k]).abs()
        assert float(diff.max()) <= 2 * 1e-3 * 3 + 1e-6, k
        assert int((diff > 2e-4).sum()) <= max(8, 0.05 * diff.numel()), (k, int((diff > 2e-4).sum()), diff.numel())


def test_fp32_full_size_golden(dev, golden_dir):
    """224x224 tile, base_channel 32: logits + argmax mask of the REAL reference (golden file)."""
    gold = np.load(os.path.join(golden_dir, "snunet_full.npz"))
    xA, xB = sar_like("full.xA", (1, 2, 224, 224)), sar_like("full.xB", (1, 2, 224, 224))
    sd = seeded_fill_(R.new_state_dict(2, 3, 32))
    m = _model(2, 32, "fp32", sd, dev).eval()
    with torch.no_grad():
        logits = m(xA.to(dev), xB.to(dev)).cpu()
    scale = float(gold["eval_logits_absmax"])
    rel = float(np.abs(logits[:, :, ::8, ::8].numpy() - gold["eval_logits_sub"]).max() / scale)
    assert rel < 1e-3, rel
    am = logits.argmax(1).numpy().astype(np.uint8)
    margin = gold["eval_margin"].astype(np.float32)
    decisive = margin > 1e-3 * scale
    mism = int((am != gold["eval_argmax"]).sum())
    assert (am[decisive] == gold["eval_argmax"][decisive]).all()
    inband = int((~decisive).sum())
    print(f"full-size: logits rel err {rel:.2e}; argmax mismatches {mism} of {am.size} (all inside the {1e-3 * scale:.1e} margin, {inband} pixels)")
    assert mism <= 8, (mism, inband, am.size)   # in-band disagreements: bounded, not just printed


def test_fp32_full_size_train_golden(dev, golden_dir):
    from kurosiwo_amd.loss import BCEandDiceLoss
    gold = np.load(os.path.join(golden_dir, "snunet_full.npz"))
    xA, xB = sar_like("full.train.xA", (2, 2, 224, 224)), sar_like("full.train.xB", (2, 2, 224, 224))
    lbl = seeded_labels("full.train.lbl", (2, 224, 224))
    sd = seeded_fill_(R.new_state_dict(2, 3, 32))
    m = _model(2, 32, "fp32", sd, dev).train()
    logits = m(xA.to(dev), xB.to(dev))
    loss = BCEandDiceLoss([1.0, 1.0, 1.0], 3, True)(logits, lbl.to(dev))
    loss.backward()
    ref = gold["train_logits_sub"]
    assert np.abs(logits.detach().cpu()[:, :, ::8, ::8].numpy() - ref).max() < 1e-3 * np.abs(ref).max()
    assert abs(float(loss) - float(gold["train_loss"])) < 1e-4 * float(gold["train_loss"])
    for k, p in m.named_parameters():
        st = gold[f"gstat.{k}"]
        nrm = float(p.grad.double().norm())
        if k.endswith("conv2.bias"):
            assert nrm < 1e-4 and st[0] < 1e-4, (k, nrm, st[0])      # analytically zero (BN follows)
            continue
        assert abs(nrm - st[0]) < 5e-3 * st[0] + 1e-6, (k, nrm, st[0])
    for k in ("conv0_0.conv1.weight", "conv_final.weight", "ca.fc1.weight", "ca1.fc2.weight", "Up1_3.up.bias"):
        g = dict(m.named_parameters())[k].grad.cpu().numpy()
        assert np.abs(g - gold[f"grad.{k}"]).max() < 5e-3 * np.abs(gold[f"grad.{k}"]).max() + 1e-7, k


@pytest.mark.parametrize("c,bc,B,H,W", [(2, 32, 2, 64, 64), (3, 32, 2, 224, 224)])      # the second: BASELINE.json configs[2] (VV, VH, DEM) at tile size
def test_bf16_train_step_close_to_oracle(dev, c, bc, B, H, W):
    from kurosiwo_amd.loss import BCEandDiceLoss
    tag = f"bf{c}{bc}{B}{H}{W}"
    xA, xB = sar_like(tag + "A", (B, c, H, W)), sar_like(tag + "B", (B, c, H, W))
    lbl = seeded_labels(tag + "L", (B, H, W))
    sd = seeded_fill_(R.new_state_dict(c, 3, bc))
    m = _model(c, bc, "bf16", sd, dev).train()
    logits = m(xA.to(dev), xB.to(dev))
    loss = BCEandDiceLoss(CLASS_WEIGHTS, 3, True)(logits, lbl.to(dev))
    loss.backward()
    ref_opt = R.AdamRef(sd, lr=0.0)
    ref_loss, ref_logits, ref_grads = R.train_step(sd, ref_opt, xA, xB, lbl, CLASS_WEIGHTS, True)
    rel = float((logits.detach().cpu() - ref_logits).abs().max() / ref_logits.abs().max())
    assert rel < 0.1, rel
    assert abs(float(loss) - ref_loss) < 0.05 * abs(ref_loss)
    cos = []
    for k, p in m.named_parameters():
        g, r = p.grad.cpu().flatten().double(), ref_grads[k].flatten().double()
        if float(r.norm()) > 1e-6:
            cos.append(float((g @ r) / (g.norm() * r.norm() + 1e-30)))
    # (two tiles per BatchNorm batch: the statistics of the deep maps -- 32 samples per channel at 64^2 -- amplify bf16 rounding, and
    # the figure moves with the rounding realisation: one-ulp differences in the first layer's bf16 outputs (a re-ordered fp32 sum,
    # tools/cfirst_ab.py) took the 64^2 median from 0.985 to 0.970 with every kernel verified equal.  bs=32 is pinned to the
    # reference golden in test_bf16_at_the_benchmarked_size_vs_reference_golden with min cosine > 0.97)
    assert np.median(cos) > 0.96, np.median(cos)
    print(f"bf16: logits rel err {rel:.3e}, median grad cosine {np.median(cos):.4f}, min {min(cos):.4f}")


def test_main_entry_end_to_end_tiny(dev, tmp_path, monkeypatch):
    """main.py with the reference's flags on a tiny synthetic set: train 1 epoch, checkpoint, reload, test."""
    import shutil
    import main as entry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copytree(os.path.join(root, "configs"), tmp_path / "configs")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("KSMI_SYNTHETIC_TILES", "8,4,4")
    miou = entry.main(["--method", "snunet", "--inputs", "pre_event_1", "post_event", "--batch_size", "4"])
    assert 0.0 <= miou <= 100.0
    ck = list((tmp_path / "checkpoints" / "snunet").glob("*/best_segmentation.pt"))
    assert ck, "best checkpoint missing"
    d = torch.load(ck[0], map_location="cpu")
    # the reference's keys (change_detection_trainer.py:206-213) + the {seed, step} words of the counter-based Dropout stream
    assert set(d) == {"epoch", "model_state_dict", "optimizer_state_dict", "lr_scheduler_state_dict", "loss", "rng_state"}
    assert len(d["model_state_dict"]) == 236


def test_headline_config_is_bitwise_reproducible_and_learns(dev):
    """BASELINE.json configs[1] at full size (bs=32, 224x224, bf16, ce+dice, Adam): size-independent properties of the fused step.
    (i) two runs from the same state give bit-identical parameters after 3 steps: every reduction has a fixed order
    (pixel-split weight-gradient slabs, BatchNorm partial rows, loss partials), nothing uses float atomics;
    (ii) the loss is finite and decreases on a repeated batch; (iii) the BatchNorm counters of the shared (siamese) encoder advance
    twice per step, once per date, as nn.BatchNorm2d does when models/snunet.py:123-131 calls the block on xA and on xB."""
    from kurosiwo_amd.snunet import SNUNet_ECAM
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    from kurosiwo_amd.trainer import CDTrainStep
    B, H, W = 32, 224, 224
    batch = make_batch(B, H, W, seed=1234)
    (xA, xB), mask = cd_inputs(batch, ("pre_event_1", "post_event"))

    def run():
        torch.manual_seed(999)
        model = SNUNet_ECAM(2, 3, base_channel=32, precision="bf16").to(dev).train()
        step = CDTrainStep(model, B, H, W, loss_function="ce+dice", lr=1e-3)
        step.set_batch(xA.to(dev), xB.to(dev), mask.to(dev))
        losses = []
        for _ in range(3):
            step.run()
            losses.append(step.loss_out.clone())
        torch.cuda.synchronize()
        nbt = model.state_dict()["conv0_0.bn1.num_batches_tracked"].item()
        return model.flat_params.clone(), torch.stack(losses).cpu(), nbt

    p1, l1, n1 = run()
    p2, l2, n2 = run()
    assert torch.equal(p1, p2) and torch.equal(l1, l2)
    assert torch.isfinite(l1).all() and float(l1[-1, 0]) < float(l1[0, 0])
    assert n1 == 6 and n2 == 6


def test_bf16_at_the_benchmarked_size_vs_reference_golden(dev, golden_dir):
    """BASELINE.json configs[1] exactly as benchmarked (batch 32, 224x224, bf16, synthetic SAR tiles) against fp32 vectors of the REAL
    reference (tests/golden/snunet_bench.npz, oracle/gen_golden.py::gen_snunet_bench): train-mode logits, ce+dice loss, argmax with a
    BOUNDED number of in-margin disagreements, gradient norms / directions, BatchNorm running statistics."""
    from kurosiwo_amd.loss import BCEandDiceLoss
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    gold = np.load(os.path.join(golden_dir, "snunet_bench.npz"))
    (xA, xB), lbl = cd_inputs(make_batch(32, 224, 224, seed=1234), ("pre_event_1", "post_event"))
    sd = seeded_fill_(R.new_state_dict(2, 3, 32))
    m = _model(2, 32, "bf16", sd, dev).train()
    logits = m(xA.to(dev), xB.to(dev))
    loss = BCEandDiceLoss([1.0, 1.0, 1.0], 3, True)(logits, lbl.to(dev))
    loss.backward()
    lg = logits.detach().float().cpu()
    scale = float(gold["train_logits_absmax"])
    err = (lg[:, :, ::16, ::16].numpy() - gold["train_logits_sub"])
    print(f"bf16 bs32: logits max err {np.abs(err).max() / scale:.4f} of scale, rms {np.sqrt((err ** 2).mean()) / scale:.5f}; "
          f"loss {float(loss):.5f} vs {float(gold['train_loss']):.5f}")
    assert np.abs(err).max() < 4e-2 * scale and np.sqrt((err ** 2).mean()) < 6e-3 * scale
    assert abs(float(loss) - float(gold["train_loss"])) < 5e-3 * float(gold["train_loss"])
    am = lg[::8].argmax(1).numpy().astype(np.uint8)
    margin = gold["train_margin_sub"].astype(np.float32)
    decisive = margin > 3e-2 * scale
    assert (am[decisive] == gold["train_argmax_sub"][decisive]).all()
    mism = int((am != gold["train_argmax_sub"]).sum())
    assert mism <= 0.01 * am.size, (mism, am.size)                     # in-margin disagreements: bounded, not just printed
    coss, bad = [], {}
    for k, p in m.named_parameters():
        st = gold[f"gstat.{k}"]
        if k.endswith("conv2.bias"):
            continue                                                   # analytically zero (BatchNorm follows)
        nrm = float(p.grad.double().norm())
        if not abs(nrm - st[0]) < 8e-2 * st[0] + 1e-6:
            bad[k] = (nrm, st[0])
        fk = f"grad.{k}"
        if fk in gold:
            g, r = p.grad.detach().double().cpu().numpy().ravel(), gold[fk].astype(np.float64).ravel()
            coss.append(float((g * r).sum() / (np.linalg.norm(g) * np.linalg.norm(r) + 1e-30)))
    assert not bad, dict(list(bad.items())[:10])
    assert min(coss) > 0.97 and sorted(coss)[1] > 0.995, coss       # (the lowest: conv0_0.bn1.weight, end of the longest backward path)
    msd = m.state_dict()
    for k in ("conv0_0.bn1", "conv0_4.bn2", "conv2_1.bn1"):
        assert np.abs(msd[f"{k}.running_mean"].cpu().numpy() - gold[f"bn.{k}.running_mean"]).max() < 2e-2 * max(1.0, np.abs(gold[f"bn.{k}.running_mean"]).max())
        assert np.abs(msd[f"{k}.running_var"].cpu().numpy() - gold[f"bn.{k}.running_var"]).max() < 3e-2 * max(1.0, float(gold[f"bn.{k}.running_var"].max()))


def test_backward_of_an_overwritten_forward_is_refused():
    """A plan keeps one set of activation buffers per (shape, mode): backward() of a forward that a later grad-enabled forward of the
    same shape has overwritten must fail loudly instead of returning the gradients of the wrong graph."""
    from kurosiwo_amd import _lib
    from kurosiwo_amd.snunet import SNUNet_ECAM
    torch.manual_seed(0)
    model = SNUNet_ECAM(2, 3, base_channel=32, precision="fp32").cuda().train()
    x = torch.randn(1, 2, 32, 32, device="cuda")
    first = model(x, x).sum()
    second = model(x, x + 1).sum()
    with pytest.raises(_lib.KsmiError, match="overwritten"):
        first.backward()
    second.backward()                       # the latest forward is still differentiable
    assert all(p.grad is not None for p in model.parameters())


@pytest.mark.parametrize("precision,B,H,W,bc", [("bf16", 4, 64, 64, 32), ("fp32", 2, 32, 48, 16), ("fp32", 8, 224, 224, 32), ("bf16", 32, 224, 224, 32)])
def test_fused_batchnorm_glue_equals_the_separate_launches(dev, precision, B, H, W, bc):
    """csrc/bnfused.hip (statistics finish inside the consuming pass: ksmi_bn_fin_add_relu with the encoder's max-pool,
    ksmi_bn_bwd_fin_apply_gated, ksmi_bnrelu_bwd_fin_apply, ksmi_bn_bwd_fin_apply_add) against the launch sequence it replaces
    (ksmi_bn_finalize / ksmi_reduce_rows + the apply passes + ksmi_maxpool2x2_forward).  The streaming arithmetic is the same
    expression.  Up to 256 statistics rows both paths sum the rows in fp64 and agree to the last bit (small cases: 0 differing words
    expected, bound 1e-6).  Longer row lists (the 224 x 224 cases) go through an in-place fold in the old path that rounds its 32
    intermediate sums to fp32, while the fused path stays in fp64: the saved statistics then differ by float ulps, which fp32
    activations carry as ~1e-6 and bf16 activations turn into occasional flips of a bf16 rounding (2^-9 each) that the 19 blocks
    spread: the bf16 full-size case is held to a statistical bound instead."""
    from kurosiwo_amd.loss import BCEandDiceLoss
    from kurosiwo_amd.snunet_plan import SNUNetPlan
    tag = f"fuse{precision}{B}{H}"
    xA, xB = sar_like(tag + "A", (B, 2, H, W)), sar_like(tag + "B", (B, 2, H, W))
    lbl = seeded_labels(tag + "L", (B, H, W))
    sd = seeded_fill_(R.new_state_dict(2, 3, bc))
    res = {}
    keep = SNUNetPlan.bn_fused
    try:
        for fused in (True, False):
            SNUNetPlan.bn_fused = fused
            m = _model(2, bc, precision, sd, dev).train()
            logits = m(xA.to(dev), xB.to(dev))
            BCEandDiceLoss(CLASS_WEIGHTS, 3, True)(logits, lbl.to(dev)).backward()
            torch.cuda.synchronize()
            plan = m.plan(B, H, W, True, True)
            names = [c[2] for c in plan.fwd.calls + plan.bwd.calls]
            assert ("ksmi_bn_fin_add_relu" in names) == fused and ("ksmi_bn_add_relu" in names) != fused
            assert ("ksmi_maxpool2x2_forward" in names) != fused
            res[fused] = (logits.detach().float().cpu().clone(), m.flat_grads.detach().cpu().clone(), m.flat_buffers.detach().cpu().clone(),
                          m.flat_counters.detach().cpu().clone())
            del m, plan, logits
            torch.cuda.empty_cache()
    finally:
        SNUNetPlan.bn_fused = keep
    (la, ga, ba, ca), (lb, gb, bb, cb) = res[True], res[False]
    assert torch.equal(ca, cb)
    ndiff = int((la != lb).sum()) + int((ga != gb).sum()) + int((ba != bb).sum())
    lmax, lrms = float((la - lb).abs().max()) / float(lb.abs().max()), float((la - lb).pow(2).mean().sqrt()) / float(lb.abs().max())
    gcos = float((ga.double() * gb.double()).sum() / (ga.double().norm() * gb.double().norm()))
    print(f"fused vs separate BatchNorm glue ({precision}, B={B}, {H}x{W}): {ndiff} differing words; logits max diff {lmax:.3g} rms {lrms:.3g} of "
          f"scale, gradients max diff {float((ga - gb).abs().max()):.3g} (scale {float(gb.abs().max()):.3g}), gradient cosine {gcos:.8f}")
    bmax = float((ba - bb).abs().max()) / float(bb.abs().max())                  # running statistics
    print(f"  running statistics max diff {bmax:.3g} of scale")
    assert bmax <= (2e-6 if H < 224 else 2e-5 if precision == "fp32" else 1e-2)
    if H < 224:
        assert lmax <= 1e-6
        # the bias gradient of conv1 is a sum over the per-workgroup rows of the apply pass: its row partition differs between the two paths
        assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max())
    elif precision == "fp32":
        assert lmax <= 2e-5 and gcos > 1 - 1e-6
    else:
        assert lmax <= 4e-2 and lrms <= 2e-3 and gcos > 0.999
