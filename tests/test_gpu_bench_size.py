"""The other headline families in the dtype they are benchmarked in, against fp32 vectors of the REAL reference
(oracle/gen_golden.py::gen_floodvit_bench / gen_changeformer_bench / gen_changeformer_bench32_eval / gen_snunet_dem_shard, run in the
build container on /root/reference):

  * BASELINE.json configs[4] per-GPU shard: FloodViT full depth (ViT d1024 L24 h16 mlp2048 + Decoder head), batch 16, bf16 (the benchmarked size);
  * BASELINE.json configs[3]: ChangeFormerV6 on 4-band SLC tiles, stochastic layers ON (counter-based stream), bf16, at batch 8
    (train-mode vectors: a batch-32 fp32 run of the reference does not fit the build container) AND at the benchmarked batch 32:
    the batch-32 plans (tile / split / ring choices depend on B) are held to reference vectors through eval mode, where a sample does
    not depend on the rest of the batch, and to size-independent properties of the train step (finite, bitwise repeatable, loss
    falls, the random stream advances);
  * BASELINE.json configs[2] per-GPU shard: SNUNet-ECAM on (VV, VH, DEM) at batch 8 (global 64 over 8 ranks), bf16 and fp32;

on the synthetic SAR tiles bench.py times (kurosiwo_amd/synthetic.make_batch, seed 1234).  SNUNet's twin of this test is
tests/test_gpu_snunet.py::test_bf16_at_the_benchmarked_size_vs_reference_golden.  Bounds = about twice what was measured on MI355X
(the measured values are printed)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]


def _cos(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


def _grad_report(model, gold, skip=()):
    """per-parameter gradient-norm ratio against the reference's statistics and direction cosine for the stored full gradients"""
    ratios, coss = {}, {}
    for k, p in model.named_parameters():
        st = gold[f"gstat.{k}"]
        if k in skip or st[0] < 1e-9 or p.grad is None:
            continue
        ratios[k] = float(p.grad.double().norm()) / float(st[0])
        if f"grad.{k}" in gold.files:
            coss[k] = _cos(p.grad.detach().float().cpu().numpy(), gold[f"grad.{k}"])
    return ratios, coss


def test_floodvit_bf16_at_the_benchmarked_size_vs_reference_golden(golden_dir):
    from kurosiwo_amd.floodvit import FinetunerSegmentation, ViT
    from kurosiwo_amd.synthetic import make_batch, seg_inputs
    from oracle import vit_ref as V
    from oracle.seeded import seeded_fill_
    gold = np.load(os.path.join(golden_dir, "floodvit_bench.npz"))
    hp = dict(channels=6, image_size=224, patch_size=16, dim=1024, depth=24, heads=16, mlp_dim=2048)
    B = 16
    enc = ViT(image_size=224, patch_size=16, num_classes=1000, dim=1024, depth=24, heads=16, mlp_dim=2048, channels=6)
    model = FinetunerSegmentation(enc, {"mlp": False, "decoder": True, "num_classes": 3, "image_size": 224, "finetuning_patch_size": 16},
                                  precision="bf16")
    sd = seeded_fill_(V.new_state_dict(**hp, head="decoder"))
    model.load_state_dict(sd)
    model = model.cuda().train()
    x, lbl = seg_inputs(make_batch(B, 224, 224, seed=1234))
    logits = model(x.cuda())
    loss = torch.nn.functional.cross_entropy(logits, lbl.cuda(), weight=torch.tensor(CLASS_WEIGHTS, device="cuda"), ignore_index=3)
    loss.backward()
    lg = logits.detach().float().cpu()
    scale = float(gold["logits_absmax"])
    err = lg[:, :, ::8, ::8].numpy() - gold["logits_sub"]
    emax, erms = float(np.abs(err).max()) / scale, float(np.sqrt((err ** 2).mean())) / scale
    plan = model.plan(B, True, True)
    tok = plan.named["feat"].float().cpu().reshape(B, 196, 1024)[::4, ::7, ::16].numpy()
    terr = float(np.abs(tok - gold["tokens_sub"]).max()) / float(gold["tokens_absmax"])
    am = lg[::4].argmax(1).numpy().astype(np.uint8)
    margin = gold["margin_sub"].astype(np.float32)
    decisive = margin > 5e-2 * scale
    mism = int((am != gold["argmax_sub"]).sum())
    ratios, coss = _grad_report(model, gold)
    r = np.array(list(ratios.values()))
    print(f"floodvit bf16 bs16: logits max err {emax:.4f} of scale, rms {erms:.5f}; encoder tokens max err {terr:.4f} of scale; loss "
          f"{float(loss):.5f} vs {float(gold['loss']):.5f}; argmax mismatches {mism} of {am.size}; gradient-norm ratio "
          f"[{r.min():.3f}, {r.max():.3f}] median {np.median(r):.4f}; cosines min {min(coss.values()):.4f} median {np.median(list(coss.values())):.4f}")
    assert emax < 3e-2 and erms < 6e-3 and terr < 3e-2                   # measured 0.0142 / 0.00285 / 0.0145
    assert abs(float(loss) - float(gold["loss"])) < 2e-3 * float(gold["loss"])          # measured 8e-4
    assert (am[decisive] == gold["argmax_sub"][decisive]).all()
    assert mism <= 0.015 * am.size, (mism, am.size)                                   # measured 0.65 %, all inside the margin band
    assert r.min() > 0.99 and r.max() < 1.01, (r.min(), r.max())                      # measured [0.997, 0.999] over all 280 parameters
    assert min(coss.values()) > 0.999, coss                                           # measured 0.9999


def test_changeformer_bf16_at_the_benchmarked_size_vs_reference_golden(golden_dir):
    from kurosiwo_amd.changeformer import ChangeFormerV6
    from kurosiwo_amd.loss import BCEandDiceLoss
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    from oracle import changeformer_ref as R
    from oracle.seeded import seeded_fill_
    gold = np.load(os.path.join(golden_dir, "changeformer_bench.npz"))
    seed, step = (int(v) for v in gold["seed_step"])
    c, B = 4, 8
    model = ChangeFormerV6(c, 3, decoder_softmax=True, embed_dim=256, precision="bf16")
    assert (model.drop_rate, model.attn_drop, model.drop_path_rate) == (0.1, 0.1, 0.1)       # changeformer.py:651-653
    model.load_state_dict(seeded_fill_(R.new_state_dict(c, 3, 256)))
    model = model.cuda().train()
    model.manual_seed(seed, step - 1)                       # the forward advances the stream: step - 1 -> step
    (x1, x2), lbl = cd_inputs(make_batch(B, 224, 224, seed=1234, channels=c), ("pre_event_1", "post_event"))
    outs = model(x1.cuda(), x2.cuda())
    assert model.rng_state().cpu().tolist() == [seed, step]
    loss = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)(outs[-1], lbl.cuda())
    loss.backward()
    emax, emean = [], []
    for i in range(4):
        e = np.abs(outs[i].detach().float().cpu().numpy() - gold[f"train.out{i}"])
        emax.append(float(e.max())); emean.append(float(e.mean()))
    e = np.abs(outs[4].detach().float().cpu()[:, :, ::8, ::8].numpy() - gold["train.out4_sub"])
    emax.append(float(e.max())); emean.append(float(e.mean()))
    am = outs[4][::2].detach().argmax(1).cpu().numpy().astype(np.uint8)
    margin = gold["train.margin_sub"].astype(np.float32)
    decisive = margin > 0.1
    mism = int((am != gold["train.argmax_sub"]).sum())
    ratios, coss = _grad_report(model, gold, skip=("TDec_x2.linear_fuse.0.bias",))      # conv bias in front of BatchNorm: analytically zero
    r = np.array(list(ratios.values()))
    print(f"changeformer bf16 bs8 SLC: sigmoid maps max err {['%.3f' % v for v in emax]} mean {['%.4f' % v for v in emean]}; loss {float(loss):.5f} vs "
          f"{float(gold['train.loss']):.5f}; argmax mismatches {mism} of {am.size} ({int((~decisive).sum())} pixels inside the margin band); "
          f"gradient-norm ratio [{r.min():.3f}, {r.max():.3f}] median {np.median(r):.4f}; cosines min {min(coss.values()):.4f} median "
          f"{np.median(list(coss.values())):.4f}")
    assert max(emean) < 1.5e-2 and max(emax) < 0.25                                   # measured 0.0076 / 0.138 (14 x 14 map, one pixel)
    assert emean[4] < 8e-3 and emax[4] < 5e-2                                         # the full-resolution map: measured 0.0039 / 0.022
    assert abs(float(loss) - float(gold["train.loss"])) < 1e-3 * float(gold["train.loss"])   # measured 1e-5
    assert (am[decisive] == gold["train.argmax_sub"][decisive]).all()
    assert mism <= 0.03 * am.size, (mism, am.size)                                    # measured 1.4 % (44 % of the pixels lie inside the band)
    assert r.min() > 0.87 and r.max() < 1.1 and 0.98 < np.median(r) < 1.02, (r.min(), r.max(), np.median(r))   # measured [0.935, 1.049], 0.9957
    assert min(coss.values()) > 0.93 and np.median(list(coss.values())) > 0.975, coss   # measured 0.964 / 0.987


def test_changeformer_batch32_eval_vs_reference_golden(golden_dir):
    """configs[3] at its stated batch: the batch-32 forward plan (bf16, SLC 4 bands) against the reference's eval outputs of the first
    8 benchmark tiles (tests/golden/changeformer_bench32_eval.npz); eval mode makes every sample independent of the other 24."""
    from kurosiwo_amd.changeformer import ChangeFormerV6
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    from oracle import changeformer_ref as R
    from oracle.seeded import seeded_fill_
    gold = np.load(os.path.join(golden_dir, "changeformer_bench32_eval.npz"))
    c, B, n = 4, 32, 8
    (x1, x2), _ = cd_inputs(make_batch(B, 224, 224, seed=1234, channels=c), ("pre_event_1", "post_event"))
    for precision, tol_mean, tol_max in (("bf16", 8e-3, 6e-2), ("fp32", 2e-5, 1e-3)):
        model = ChangeFormerV6(c, 3, decoder_softmax=True, embed_dim=256, precision=precision)
        model.load_state_dict(seeded_fill_(R.new_state_dict(c, 3, 256)))
        model = model.cuda().eval()
        with torch.no_grad():
            outs = model(x1.cuda(), x2.cuda())
        emax, emean = [], []
        for i in range(4):
            e = np.abs(outs[i][:n].float().cpu().numpy() - gold[f"eval.out{i}"])
            emax.append(float(e.max())); emean.append(float(e.mean()))
        e = np.abs(outs[4][:n, :, ::8, ::8].float().cpu().numpy() - gold["eval.out4_sub"])
        emax.append(float(e.max())); emean.append(float(e.mean()))
        am = outs[4][:n:2].argmax(1).cpu().numpy().astype(np.uint8)
        margin = gold["eval.margin_sub"].astype(np.float32)
        decisive = margin > (0.1 if precision == "bf16" else 2e-3)
        mism = int((am != gold["eval.argmax_sub"]).sum())
        print(f"changeformer {precision} bs32 eval, first 8 tiles: sigmoid maps max err {['%.4f' % v for v in emax]} mean {['%.5f' % v for v in emean]}; "
              f"argmax mismatches {mism} of {am.size} ({int((~decisive).sum())} pixels inside the margin band)")
        assert max(emean) < tol_mean and max(emax) < (0.3 if precision == "bf16" else tol_max), (emean, emax)
        assert emax[4] < tol_max, emax
        assert (am[decisive] == gold["eval.argmax_sub"][decisive]).all()
        assert mism <= (0.03 if precision == "bf16" else 1e-4) * am.size, (mism, am.size)
        del model, outs
        torch.cuda.empty_cache()


def test_changeformer_batch32_train_step_properties():
    """configs[3] exactly as bench.py --model changeformer runs it (bs 32, SLC 4 bands, bf16, SGD 6e-4 / 0.99 / 1e-5, ce+dice, stochastic
    layers on): the train step is finite, bit-for-bit repeatable from the same state (every reduction has a fixed order, the Bernoulli
    draws come from the counter-based stream), the loss falls on a repeated batch, the random stream advances once per step."""
    from kurosiwo_amd.changeformer import ChangeFormerV6
    from kurosiwo_amd.optim import FusedSGD
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    from kurosiwo_amd.trainer import CDTrainStep
    c, B = 4, 32
    (x1, x2), lbl = cd_inputs(make_batch(B, 224, 224, seed=1234, channels=c), ("pre_event_1", "post_event"))

    def run():
        torch.manual_seed(999)
        model = ChangeFormerV6(c, 3, decoder_softmax=True, embed_dim=256, precision="bf16").cuda().train()
        model.manual_seed(77, 0)
        opt = FusedSGD(model.parameters(), lr=6e-4, momentum=0.99, weight_decay=1e-5)
        step = CDTrainStep(model, B, 224, 224, loss_function="ce+dice", optimizer=opt, bucket_mb=16.0)
        step.set_batch(x1.cuda(), x2.cuda(), lbl.cuda())
        losses = []
        for _ in range(4):
            step.run()
            losses.append(step.loss_out.clone())
        torch.cuda.synchronize()
        words = model.rng_state().cpu().tolist()
        res = model.flat_params.clone(), model.flat_grads.clone(), torch.stack(losses).cpu(), words
        del step, model
        torch.cuda.empty_cache()
        return res

    p1, g1, l1, w1 = run()
    p2, g2, l2, w2 = run()
    print("changeformer bs32 train losses", [round(float(v), 5) for v in l1[:, 0]], "rng words", w1)
    assert torch.isfinite(l1).all() and torch.isfinite(p1).all() and torch.isfinite(g1).all()
    assert torch.equal(p1, p2) and torch.equal(g1, g2) and torch.equal(l1, l2)
    assert float(l1[-1, 0]) < float(l1[0, 0])
    assert w1 == [77, 4] and w2 == w1
    assert float(g1.abs().max()) > 0


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_snunet_dem_shard_vs_reference_golden(golden_dir, precision):
    """BASELINE.json configs[2] per-GPU shard (3 channels per date: VV, VH, DEM; batch 8 = global 64 over 8 ranks; 224 x 224) against
    fp32 vectors of the REAL reference (tests/golden/snunet_dem_shard.npz): train-mode logits, ce+dice loss, argmax with the number of
    in-margin disagreements printed and bounded, gradient norms and directions.  The DEM plane goes in through forward(xA, xB, dem)
    (the shared tail channels) as well as concatenated."""
    from kurosiwo_amd.loss import BCEandDiceLoss
    from kurosiwo_amd.snunet import SNUNet_ECAM
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    from oracle import snunet_ref as R
    from oracle.seeded import seeded_fill_, seeded_tensor
    gold = np.load(os.path.join(golden_dir, "snunet_dem_shard.npz"))
    B = 8
    (xA, xB), lbl = cd_inputs(make_batch(B, 224, 224, seed=4321), ("pre_event_1", "post_event"))
    dem = torch.nn.functional.interpolate(seeded_tensor("snunet_dem_shard.dem", (B, 1, 14, 14)), size=(224, 224), mode="bilinear", align_corners=False)
    sd = seeded_fill_(R.new_state_dict(3, 3, 32))
    m = SNUNet_ECAM(3, 3, base_channel=32, precision=precision)
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m = m.cuda().train()
    logits = m(xA.cuda(), xB.cuda(), dem.cuda())
    loss = BCEandDiceLoss([1.0, 1.0, 1.0], 3, True)(logits, lbl.cuda())
    loss.backward()
    lg = logits.detach().float().cpu()
    scale = float(gold["train_logits_absmax"])
    err = lg[:, :, ::8, ::8].numpy() - gold["train_logits_sub"]
    emax, erms = float(np.abs(err).max()) / scale, float(np.sqrt((err ** 2).mean())) / scale
    am = lg[::2].argmax(1).numpy().astype(np.uint8)
    margin = gold["train_margin_sub"].astype(np.float32)
    band = (3e-2 if precision == "bf16" else 1e-3) * scale
    decisive = margin > band
    mism = int((am != gold["train_argmax_sub"]).sum())
    print(f"snunet c=3 bs8 {precision}: logits max err {emax:.2e} of scale, rms {erms:.2e}; loss {float(loss):.6f} vs {float(gold['train_loss']):.6f}; "
          f"argmax mismatches {mism} of {am.size} ({int((~decisive).sum())} pixels inside the {band:.3g} margin band)")
    if precision == "fp32":
        assert emax < 1e-3, emax                                   # north-star: 1e-3 rel on logits
        assert abs(float(loss) - float(gold["train_loss"])) < 1e-4 * float(gold["train_loss"])
        assert mism <= 8, mism                                     # exact outside the band; bounded inside
    else:
        assert emax < 4e-2 and erms < 6e-3, (emax, erms)
        assert abs(float(loss) - float(gold["train_loss"])) < 5e-3 * float(gold["train_loss"])
        assert mism <= 0.01 * am.size, (mism, am.size)
    assert (am[decisive] == gold["train_argmax_sub"][decisive]).all()
    bad, coss = {}, []
    # bf16: held to the SAME step of the imported reference with bf16 storage emulated (oracle/bf16_storage.py ->
    # snunet_dem_shard_bf16emu.npz, oracle/gen_bf16emu_golden.py): bf16 storage lowers the first block's gradient norms by 5-9 % on the
    # reference's own module graph (forward rounding alone does it: LABNOTES.md round 5), so the fp32 vectors would only bound the bf16
    # path loosely (round 4: 16 %).  Against the emulated reference every norm is within 8 % -- two rounding realisations of one
    # arithmetic contract (measured: printed below) -- and the deficit against fp32 must be the emulated reference's, not a larger one.
    emu = np.load(os.path.join(golden_dir, "snunet_dem_shard_bf16emu.npz")) if precision == "bf16" else None
    tol = 8e-2 if precision == "bf16" else 5e-3
    ratios = {}
    for k, p in m.named_parameters():
        st = gold[f"gstat.{k}"]
        if k.endswith("conv2.bias"):
            continue                                               # analytically zero (BatchNorm follows)
        nrm = float(p.grad.double().norm())
        ref = float(emu[f"gstat.{k}"][0]) if emu is not None else float(st[0])
        ratios[k] = nrm / max(ref, 1e-30)
        if not abs(nrm - ref) < tol * ref + 1e-6:
            bad[k] = (nrm, ref, float(st[0]))
        if f"grad.{k}" in gold.files:
            coss.append(_cos(p.grad.detach().float().cpu().numpy(), gold[f"grad.{k}"]))
    if emu is not None:
        rs = np.array(list(ratios.values()))
        lo, hi = min(ratios, key=ratios.get), max(ratios, key=ratios.get)
        print(f"bf16 gradient norms / emulated-reference norms over {rs.size} parameters: median {np.median(rs):.4f}, min {rs.min():.4f} ({lo}), "
              f"max {rs.max():.4f} ({hi}); first block vs fp32: HIP {float(m.conv0_0.conv1.weight.grad.double().norm()) / float(gold['gstat.conv0_0.conv1.weight'][0]):.4f}, "
              f"emulated reference {float(emu['gstat.conv0_0.conv1.weight'][0]) / float(gold['gstat.conv0_0.conv1.weight'][0]):.4f}; "
              f"loss {float(loss):.6f} vs emulated {float(emu['train_loss']):.6f} vs fp32 {float(gold['train_loss']):.6f}")
        assert abs(np.median(rs) - 1.0) < 1e-2, np.median(rs)          # no systematic offset against the emulated reference
        assert abs(float(loss) - float(emu["train_loss"])) < 2e-3 * float(emu["train_loss"])
        ecos = [_cos(m.get_parameter(k).grad.detach().float().cpu().numpy(), emu[f"grad.{k}"]) for k in
                ("conv0_0.conv1.weight", "conv0_0.conv2.weight", "conv0_4.conv2.weight", "conv_final.weight")]
        print("cosine with the emulated reference's gradients (first block conv1, conv2, last block conv2, head):", np.round(ecos, 4).tolist())
        assert min(ecos) > 0.93, ecos
    assert not bad, dict(list(bad.items())[:10])
    assert min(coss) > (0.95 if precision == "bf16" else 0.9999), coss      # (bf16: the first convolution's direction, same realisation spread)
    # the concatenated form of the same inputs is the same function
    if precision == "fp32":
        m.zero_grad()
        with torch.no_grad():
            m.eval()
            a = m(xA.cuda(), xB.cuda(), dem.cuda())
            b = m(torch.cat((xA, dem), 1).cuda(), torch.cat((xB, dem), 1).cuda())
        assert torch.equal(a, b)
