"""GPU parity of the ChangeFormerV6 path (kurosiwo_amd/changeformer.py) against the CPU oracle (oracle/changeformer_ref.py)
and the golden vectors generated from the real reference (tests/golden/changeformer.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]


def sar_like(name, shape):
    from oracle.seeded import seeded_tensor
    return seeded_tensor(name, shape).clamp_(-2.23, 5.75)


def build(precision, stochastic=False):
    from kurosiwo_amd.changeformer import ChangeFormerV6
    from oracle import changeformer_ref as R
    from oracle.seeded import seeded_fill_
    model = ChangeFormerV6(2, 3, decoder_softmax=True, embed_dim=256, precision=precision)
    assert (model.drop_rate, model.attn_drop, model.drop_path_rate) == (0.1, 0.1, 0.1)       # changeformer.py:651-653
    if not stochastic:      # the p = 0 golden vectors: train mode = BatchNorm batch statistics only
        model.drop_rate = model.attn_drop = model.drop_path_rate = 0.0
    sd = seeded_fill_(R.new_state_dict(2, 3, 256))
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    return model.cuda(), sd


def relerr(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


def tok(t, B):          # [2B*hw, C] tokens -> date-1 half as [B, hw, C]
    t = t.float().cpu()
    return t.reshape(2 * B, -1, t.shape[-1])[:B]


def nchw(t, B, h, w):
    return t.float().cpu().reshape(B, h, w, -1).permute(0, 3, 1, 2)


def compare_intermediates(plan, inter, B, tol):
    errs = {}
    for st in range(4):
        errs[f"pe{st + 1}"] = relerr(tok(plan.named[f"pe{st + 1}"], B), inter[f"A.pe{st + 1}"])
        last = f"s{st + 1}b{[3, 3, 4, 3][st] - 1}"
        errs[last] = relerr(tok(plan.named[last], B), inter[f"A.{last}"])
        h = 56 >> st
        errs[f"f{st + 1}"] = relerr(nchw(plan.named[f"f{st + 1}"][:B * h * h], B, h, h), inter[f"A.f{st + 1}"])
        errs[f"fB{st + 1}"] = relerr(nchw(plan.named[f"f{st + 1}"][B * h * h:], B, h, h), inter[f"B.f{st + 1}"])
    for i, h in ((4, 7), (3, 14), (2, 28), (1, 56)):
        errs[f"c{i}"] = relerr(nchw(plan.named[f"c{i}"], B, h, h), inter[f"c{i}"])
    errs["fuse"] = relerr(nchw(plan.named["fuse"], B, 56, 56), inter["fuse"])
    errs["dense_2x"] = relerr(nchw(plan.named["dense_2x"], B, 112, 112), inter["dense_2x"])
    errs["dense_1x"] = relerr(nchw(plan.named["dense_1x"], B, 224, 224), inter["dense_1x"])
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, errs


def test_eval_forward_vs_oracle_and_golden(golden_dir):
    from oracle import changeformer_ref as R
    gold = np.load(os.path.join(golden_dir, "changeformer.npz"))
    model, sd = build("fp32")
    model.eval()
    x1 = sar_like("changeformer.eval.x1", (1, 2, 224, 224))
    x2 = sar_like("changeformer.eval.x2", (1, 2, 224, 224))
    inter = {}
    with torch.no_grad():
        ref = R.changeformer_forward(sd, x1, x2, training=False, inter=inter)
        outs = model(x1.cuda(), x2.cuda())
    compare_intermediates(model.plan(1, 224, 224, False, False), inter, 1, 5e-4)
    assert [tuple(o.shape) for o in outs] == [(1, 3, 7, 7), (1, 3, 14, 14), (1, 3, 28, 28), (1, 3, 56, 56), (1, 3, 224, 224)]
    for i in range(5):
        assert float((outs[i].cpu() - ref[i]).abs().max()) < 1e-3            # north_star: 1e-3 on the outputs (sigmoid maps)
    for i in range(4):
        assert np.abs(outs[i].cpu().numpy() - gold[f"eval.out{i}"]).max() < 1e-3
    assert np.abs(outs[4].cpu()[:, :, ::8, ::8].numpy() - gold["eval.out4_sub"]).max() < 1e-3
    confident = gold["eval.margin"].astype(np.float32) > 2e-3
    am = outs[4].argmax(1).cpu().numpy().astype(np.uint8)
    assert (am == gold["eval.argmax"])[confident].all()
    mism, inband = int((am != gold["eval.argmax"]).sum()), int((~confident).sum())
    print(f"changeformer eval argmax: {mism} mismatches of {am.size}, all among the {inband} pixels inside the 2e-3 margin")
    assert mism <= 8, (mism, inband, am.size)   # bounded, not just excluded


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_train_forward_backward_vs_oracle(golden_dir, precision):
    from oracle import changeformer_ref as R
    from oracle.seeded import seeded_labels
    gold = np.load(os.path.join(golden_dir, "changeformer.npz"))
    B = 2
    model, sd = build(precision)
    model.train()
    x1 = sar_like("changeformer.train.x1", (B, 2, 224, 224))
    x2 = sar_like("changeformer.train.x2", (B, 2, 224, 224))
    lbl = seeded_labels("changeformer.train.lbl", (B, 224, 224))
    outs = model(x1.cuda(), x2.cuda())
    plan = model.plan(B, 224, 224, True, True)
    inter = {}
    with torch.no_grad():
        ref = R.changeformer_forward(sd, x1, x2, training=True, inter=inter)
    compare_intermediates(plan, inter, B, 5e-4 if precision == "fp32" else 0.15)
    for i in range(5):
        err = (outs[i].detach().cpu() - ref[i]).abs()
        if precision == "fp32":
            assert float(err.max()) < 1e-3, i
        else:       # bf16 storage through 13 blocks + BatchNorm over as few as 98 pixels: bound the mean, sanity-bound the max
            # (the maximum over 3 x 10^5 probabilities moves with the summation order of the GEMMs: 0.21-0.26 observed)
            q = float(torch.quantile(err.flatten()[::7].float(), 0.999))
            print(f"bf16 train step out{i}: mean {float(err.mean()):.4f} q999 {q:.4f} max {float(err.max()):.4f}")
            # measured on MI355X: mean <= 0.0103, q999 <= 0.0585, max <= 0.21 (one 14 x 14 pixel)
            assert float(err.mean()) < 2e-2 and q < 0.09 and float(err.max()) < 0.35, (i, float(err.mean()), q, float(err.max()))
    if precision == "fp32":
        assert np.abs(outs[4].detach().cpu()[:, :, ::8, ::8].numpy() - gold["train.out4_sub"]).max() < 1e-3
    # BatchNorm running statistics after the step
    msd = model.state_dict()
    for k in ("TDec_x2.diff_c4.2", "TDec_x2.diff_c1.2", "TDec_x2.make_pred_c2.2", "TDec_x2.linear_fuse.1"):
        rt = 1e-3 if precision == "fp32" else 5e-2
        assert np.abs(msd[f"{k}.running_mean"].cpu().numpy() - gold[f"bn.{k}.running_mean"]).max() < rt * max(1.0, np.abs(gold[f"bn.{k}.running_mean"]).max())
        assert np.abs(msd[f"{k}.running_var"].cpu().numpy() - gold[f"bn.{k}.running_var"]).max() < rt * max(1.0, float(gold[f"bn.{k}.running_var"].max()))
        assert int(msd[f"{k}.num_batches_tracked"]) == 1
    # backward on output[-1] with the reference's CD criterion; the oracle uses the GPU's ReLU active sets
    from kurosiwo_amd.loss import BCEandDiceLoss
    crit = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)
    loss = crit(outs[-1], lbl.cuda())
    loss.backward()
    masks = {}
    for i, h in ((4, 7), (3, 14), (2, 28), (1, 56)):
        sc = plan.scales[i]
        masks[f"diff_c{i}.0"] = (nchw(sc["r1"], B, h, h) > 0).float()
        masks[f"diff_c{i}.3"] = (nchw(sc["r2"], B, h, h) > 0).float()
    masks["dense_2x"] = (nchw(plan.dec["Ra"], B, 112, 112) > 0).float()
    masks["dense_1x"] = (nchw(plan.dec["Rb"], B, 224, 224) > 0).float()
    _, ref_loss, ref_grads, _ = R.loss_and_grads(sd, x1, x2, lbl, CLASS_WEIGHTS, True, masks=masks)
    assert abs(float(loss) - ref_loss) < (2e-4 if precision == "fp32" else 3e-2)
    if precision == "fp32":
        assert abs(float(loss) - float(gold["train.loss"])) < 2e-4
    worst, coss = {}, []
    for k, p in model.named_parameters():
        g, r = p.grad.detach().float().cpu(), ref_grads[k]
        if float(r.abs().max()) == 0.0:
            assert float(g.abs().max()) == 0.0, k
            continue
        if k == "TDec_x2.linear_fuse.0.bias":
            # a conv bias followed directly by BatchNorm has an analytically zero gradient: both sides hold rounding noise
            assert float(g.abs().max()) < (1e-5 if precision == "fp32" else 1e-2), k
            continue
        if precision == "fp32":
            e = float((g - r).abs().max() / (r.abs().max() + 1e-12))
            l2 = float((g - r).double().norm() / (r.double().norm() + 1e-30))
            if not (l2 < 2e-3 and e < 5e-3):
                worst[k] = (e, l2)
        else:
            cos = float((g.double() * r.double()).sum() / (g.double().norm() * r.double().norm() + 1e-30))
            coss.append(cos)
            if not cos > 0.85:          # BatchNorm over 98 pixels at the 7x7 scale makes single bf16 gradients noisy: see the quantile bound below
                worst[k] = cos
    if coss:
        assert float(np.median(coss)) > 0.97, float(np.median(coss))
        # (0.95-0.98 observed: the stage-4 gradients -- BatchNorm over 98 pixels -- move by tens of percent with the fp32 summation
        # order of the token GEMMs, e.g. gemm.hip vs gemm2.hip, whose outputs agree to the last bf16 digit at op level)
        assert float(np.mean(np.array(coss) > 0.95)) > 0.93, float(np.mean(np.array(coss) > 0.95))
    assert not worst, f"{precision}: {len(worst)} params: {dict(list(worst.items())[:12])}"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_train_step_with_stochastic_layers_vs_reference_golden(golden_dir, precision):
    """Dropout(0.1) / attention dropout 0.1 / DropPath(linspace(0, 0.1, 13)) ON (changeformer.py:651-653).  The golden vector is the
    reference's own module graph with the draws of its nn.Dropout / DropPath instances taken from the counter-based stream
    (oracle/gen_golden.py:gen_changeformer_drop); the HIP kernels regenerate the same masks in the forward and the backward pass."""
    from oracle import changeformer_ref as R
    from oracle import rng_ref as G
    from oracle.seeded import seeded_labels
    from kurosiwo_amd.loss import BCEandDiceLoss
    gold = np.load(os.path.join(golden_dir, "changeformer_drop.npz"))
    seed, step = (int(v) for v in gold["seed_step"])
    B = 2
    model, sd = build(precision, stochastic=True)
    model.train()
    model.manual_seed(seed, step - 1)                       # the forward advances the stream: step - 1 -> step
    x1 = sar_like("changeformer.drop.x1", (B, 2, 224, 224))
    x2 = sar_like("changeformer.drop.x2", (B, 2, 224, 224))
    lbl = seeded_labels("changeformer.drop.lbl", (B, 224, 224))
    outs = model(x1.cuda(), x2.cuda())
    assert model.rng_state().cpu().tolist() == [seed, step]
    plan = model.plan(B, 224, 224, True, True)
    stream = G.DropStream(seed, step)
    inter = {}
    with torch.no_grad():
        ref = R.changeformer_forward(sd, x1, x2, training=True, inter=inter, stream=stream)
    compare_intermediates(plan, inter, B, 5e-4 if precision == "fp32" else 0.15)
    if precision == "fp32":
        for i in range(4):
            assert np.abs(outs[i].detach().cpu().numpy() - gold[f"train.out{i}"]).max() < 1e-3
        assert np.abs(outs[4].detach().cpu()[:, :, ::8, ::8].numpy() - gold["train.out4_sub"]).max() < 1e-3
        for i in range(5):
            assert float((outs[i].detach().cpu() - ref[i]).abs().max()) < 1e-3
    else:
        for i in range(5):
            err = (outs[i].detach().cpu() - ref[i]).abs()
            print(f"bf16 stochastic step out{i}: mean {float(err.mean()):.4f} max {float(err.max()):.4f}")
            assert float(err.mean()) < 1.7e-2 and float(err.max()) < 0.16, (i, float(err.mean()), float(err.max()))     # measured <= 0.0085 / 0.079
    crit = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)
    loss = crit(outs[-1], lbl.cuda())
    loss.backward()
    assert abs(float(loss) - float(gold["train.loss"])) < (2e-4 if precision == "fp32" else 3e-2)
    masks = {}
    for i, h in ((4, 7), (3, 14), (2, 28), (1, 56)):
        sc = plan.scales[i]
        masks[f"diff_c{i}.0"] = (nchw(sc["r1"], B, h, h) > 0).float()
        masks[f"diff_c{i}.3"] = (nchw(sc["r2"], B, h, h) > 0).float()
    masks["dense_2x"] = (nchw(plan.dec["Ra"], B, 112, 112) > 0).float()
    masks["dense_1x"] = (nchw(plan.dec["Rb"], B, 224, 224) > 0).float()
    _, ref_loss, ref_grads, _ = R.loss_and_grads(sd, x1, x2, lbl, CLASS_WEIGHTS, True, masks=masks, stream=stream)
    worst, coss = {}, []
    for k, p in model.named_parameters():
        g, r = p.grad.detach().float().cpu(), ref_grads[k]
        if float(r.abs().max()) == 0.0:
            assert float(g.abs().max()) == 0.0, k
            continue
        if k == "TDec_x2.linear_fuse.0.bias":
            continue
        if precision == "fp32":
            e = float((g - r).abs().max() / (r.abs().max() + 1e-12))
            l2 = float((g - r).double().norm() / (r.double().norm() + 1e-30))
            if not (l2 < 2e-3 and e < 5e-3):
                worst[k] = (e, l2)
            ref = gold[f"gstat.{k}"]
            if not abs(float(g.double().norm()) - ref[0]) <= 5e-3 * ref[0] + 1e-7:        # the reference's own gradient norms
                worst[k + " (golden norm)"] = (float(g.double().norm()), ref[0])
        else:
            cos = float((g.double() * r.double()).sum() / (g.double().norm() * r.double().norm() + 1e-30))
            coss.append(cos)
            if not cos > 0.85:
                worst[k] = cos
    if coss:
        assert float(np.median(coss)) > 0.97, float(np.median(coss))
    assert not worst, f"{precision}: {len(worst)} params: {dict(list(worst.items())[:12])}"
    # a second step draws new masks (step + 1): the output must change; the same (seed, step) must reproduce the first step bit for bit
    model.zero_grad(set_to_none=True)
    first = outs[4].detach().clone()
    second = model(x1.cuda(), x2.cuda())[4].detach()
    assert float((second - first).abs().max()) > 1e-4
    model2, _ = build(precision, stochastic=True)
    model2.train()
    model2.manual_seed(seed, step - 1)
    again = model2(x1.cuda(), x2.cuda())[4].detach()
    assert torch.equal(again, first)


def test_main_entry_changeformer_end_to_end_tiny(tmp_path, monkeypatch):
    """main.py --method changeformer on a tiny synthetic set: SGD(momentum .99, wd 1e-5) epoch, checkpoint, reload, test."""
    import shutil
    import main as entry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copytree(os.path.join(root, "configs"), tmp_path / "configs")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("KSMI_SYNTHETIC_TILES", "8,4,4")
    miou = entry.main(["--method", "changeformer", "--inputs", "pre_event_1", "post_event", "--batch_size", "4"])
    assert 0.0 <= miou <= 100.0
    ck = list((tmp_path / "checkpoints" / "changeformer").glob("*/best_segmentation.pt"))
    assert ck, "best checkpoint missing"
    d = torch.load(ck[0], map_location="cpu")
    assert len(d["model_state_dict"]) == 373


def test_main_entry_multi_scale_infer(tmp_path, monkeypatch):
    """configs/method/changeformer: multi_scale_infer = true (change_detection_trainer.py:139-146): the train-time metric predictions are
    the mean of the five outputs (kurosiwo_amd/training/change_detection_trainer.py: multi_scale_prediction); losses, weights and the
    evaluation loop (which always takes output[-1], :394-395) do not change -> the same validation mIoU as the default run."""
    import re
    import shutil
    import main as entry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for flag in (False, True):
        wd = tmp_path / ("ms" if flag else "plain")
        wd.mkdir()
        shutil.copytree(os.path.join(root, "configs"), wd / "configs")
        cfg = wd / "configs" / "method" / "changeformer" / "changeformer.json"
        if flag:
            txt = re.sub(r'"multi_scale_infer"\s*:\s*false', '"multi_scale_infer": true', cfg.read_text())
            assert '"multi_scale_infer": true' in txt
            cfg.write_text(txt)
        monkeypatch.chdir(wd)
        monkeypatch.setenv("KSMI_SYNTHETIC_TILES", "8,4,4")
        res.append(entry.main(["--method", "changeformer", "--inputs", "pre_event_1", "post_event", "--batch_size", "4"]))
    assert res[0] == res[1], res


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_slc_four_band_inputs_vs_reference_golden(golden_dir, precision):
    """BASELINE.json configs[3] as written: SLC tiles, 4 bands per date -> ChangeFormerV6(input_nc=4).  HIP path against the golden
    vectors of the REAL reference (tests/golden/changeformer_slc.npz): eval probability maps + argmax, train loss, gradients."""
    from kurosiwo_amd.changeformer import ChangeFormerV6
    from kurosiwo_amd.loss import BCEandDiceLoss
    from oracle import changeformer_ref as R
    from oracle.seeded import seeded_fill_, seeded_labels
    gold = np.load(os.path.join(golden_dir, "changeformer_slc.npz"))
    model = ChangeFormerV6(4, 3, decoder_softmax=True, embed_dim=256, precision=precision)
    model.drop_rate = model.attn_drop = model.drop_path_rate = 0.0          # the golden train step has the stochastic layers at p = 0
    sd = seeded_fill_(R.new_state_dict(4, 3, 256))
    assert list(model.state_dict().keys()) == list(gold["state_dict_keys"])
    model.load_state_dict(sd)
    model = model.cuda().eval()
    x1 = sar_like("changeformer.slc.eval.x1", (1, 4, 224, 224))
    x2 = sar_like("changeformer.slc.eval.x2", (1, 4, 224, 224))
    with torch.no_grad():
        outs = model(x1.cuda(), x2.cuda())
    def close(a, b, what):
        e = np.abs(a - b)
        if precision == "fp32":
            assert e.max() < 1e-3, (what, e.max())
        else:                       # bf16 storage through 13 blocks: bound the mean tightly, the maximum loosely
            print(f"bf16 slc eval {what}: mean {e.mean():.4f} max {e.max():.4f}")
            assert e.mean() < 8e-3 and e.max() < 0.11, (what, e.mean(), e.max())            # measured <= 0.0039 / 0.054
    for i in range(4):
        close(outs[i].cpu().numpy(), gold[f"eval.out{i}"], i)
    close(outs[4].cpu()[:, :, ::8, ::8].numpy(), gold["eval.out4_sub"], 4)
    margin = gold["eval.margin"].astype(np.float32)
    band = 2e-3 if precision == "fp32" else 0.06                                            # measured: no flip above a margin of 0.029
    am = outs[4].argmax(1).cpu().numpy().astype(np.uint8)
    assert (am == gold["eval.argmax"])[margin > band].all()
    inband = int((am != gold["eval.argmax"]).sum())
    mm = margin[am != gold["eval.argmax"]]
    print(f"slc eval argmax: {inband} flips of {am.size}, largest margin among them {float(mm.max()) if mm.size else 0.0:.4f}")
    assert inband <= (50 if precision == "fp32" else 700), inband      # bounded, not just printed (bf16 measured 340 of 50176)
    # train step: loss + gradients (stochastic layers at p = 0 on both sides)
    model.train()
    x1 = sar_like("changeformer.slc.train.x1", (2, 4, 224, 224))
    x2 = sar_like("changeformer.slc.train.x2", (2, 4, 224, 224))
    lbl = seeded_labels("changeformer.slc.train.lbl", (2, 224, 224))
    outs = model(x1.cuda(), x2.cuda())
    loss = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)(outs[-1], lbl.cuda())
    loss.backward()
    print(f"slc train: loss diff {abs(float(loss) - float(gold['train.loss'])):.5f} out4 max "
          f"{np.abs(outs[4].detach().cpu()[:, :, ::8, ::8].numpy() - gold['train.out4_sub']).max():.4f}")
    assert abs(float(loss) - float(gold["train.loss"])) < (3e-4 if precision == "fp32" else 1e-3)               # bf16 measured 2e-5
    assert np.abs(outs[4].detach().cpu()[:, :, ::8, ::8].numpy() - gold["train.out4_sub"]).max() < (1e-3 if precision == "fp32" else 0.07)   # 0.035
    k = "Tenc_x2.patch_embed1.proj.weight"
    g = dict(model.named_parameters())[k].grad.detach().float().cpu().numpy()
    assert g.shape == (64, 4, 7, 7)
    ref = gold[f"grad.{k}"]
    cos = float((g.astype(np.float64) * ref).sum() / (np.linalg.norm(g.astype(np.float64)) * np.linalg.norm(ref.astype(np.float64)) + 1e-30))
    print(f"slc patch-embed grad cosine {cos:.5f}")
    assert cos > (0.999 if precision == "fp32" else 0.95), cos                                                   # bf16 measured 0.974
    if precision == "fp32":
        bad = {}
        for kk, p in model.named_parameters():
            st = gold[f"gstat.{kk}"]
            nrm = float(p.grad.double().norm()) if p.grad is not None else 0.0
            if kk == "TDec_x2.linear_fuse.0.bias":
                continue                      # conv bias in front of BatchNorm: analytically zero
            if abs(nrm - st[0]) > 2e-2 * st[0] + 1e-6:
                bad[kk] = (nrm, st[0])
        assert not bad, dict(list(bad.items())[:8])
