"""The bf16 parity gate of SURVEY.md §8(d)(ii): after an identical K-step training run from identical weights, the mIoU of the HIP
path on the 64 held-out synthetic tiles (seed 424242) is within +-0.002 of the CPU fp32 run.

The CPU side is a run of the IMPORTED REFERENCE (oracle/gen_parity_run.py --reference -> tests/golden/snunet_parity_run_ref.npz:
/root/reference/models/snunet.py + utilities/bce_and_dice.py + torch.optim.Adam driven as change_detection_trainer.py:135-180 does,
40 Adam steps of ce+dice on batches of 4 tiles, eval-mode inference on the held-out tiles after 20 and after 40 steps; minutes on 8
CPU threads, so it is a committed fixture rather than recomputed here).  The oracle's own run of the protocol
(snunet_parity_run.npz) is held to it in tests/test_oracle_snunet.py (mIoU within 1e-5 at both checkpoints).  The HIP side repeats the protocol through the fused train step in bf16 (the
benchmarked dtype) and in fp32.

What is asserted, and why two checkpoints (round 5: on EVERY draw of the HIP run, see DRAWS below).  K = 40 is on the plateau of the learning curve (mIoU 0.986): there the gate is the
survey's +-0.002 for bf16 (fp32: 5e-4).  K = 20 is on the steep part (mIoU rises 0.66 -> 0.97 between steps 10 and 20): fp32 HIP
still tracks the CPU run to 2e-4, while a bf16 TRAINING trajectory is a slightly different trajectory and sits up to 0.015 lower
at that step before it rejoins (measured: -0.0144 at 20, +0.0011 at 40, -0.0005 at 80); bf16 INFERENCE is not the cause -- evaluating
the same weights in bf16 and in fp32 differs by < 5e-5 mIoU (last assertion).  The K = 20 bf16 bound (0.03) records that behaviour.
Reference semantics: training/change_detection_trainer.py:135-180 (step), :152,189 (argmax, mIoU = IoU[:3].mean())."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


# Draws of the HIP run: the seeded weights and the same weights with conv0_0.conv1.weight scaled by 1 + p (float32: p acts in units of
# 1.19e-7 above 1 and 6e-8 below).  EVERY draw is asserted -- no median-only bound, no dropped sample.
#
# Round 5 found why round 4 needed one: the convolution epilogues summed the BatchNorm statistics over their fp32 accumulators while
# BatchNorm normalised the bf16 values they STORED.  Where the rounding noise of a channel is comparable with its spread the
# normalised values then have variance > 1, and a 40-step Adam run took loss excursions of 2.6 x ... 16 x in about one draw in
# eleven (profiles/r04_parity_draws.txt; K = 40 off by -0.033 on the seeded weights).  With the statistics taken over the stored
# values (what the reference's BatchNorm sees) the eleven draws are K = 40: -0.0002 ... +0.0019, K = 20: -0.0114 ... -0.0010, loss
# within 0.47 of the fp32 trajectory (profiles/r05_parity_draws.txt).
#
# What the bounds are held to: tests/golden/snunet_parity_draws_ref.npz = the same protocol on the IMPORTED reference under 23 such
# perturbations with bf16 STORAGE emulated (oracle/bf16_storage.py, oracle/gen_parity_draws.py) and 4 in fp32.  The reference's own fp32
# run moves by <= 5e-4 under a 1e-7 perturbation; its bf16-storage run by -0.040 ... +0.0003 at K = 20 (still on the steep part of the
# learning curve) and -0.0036 ... +0.0016 at K = 40 (3 draws of 23 outside the survey's +-0.002), loss up to 2.8 x the fp32 trajectory
# at step 5 (2 excursions above 1.0).  Sixty further HIP draws (profiles/r05_parity_draws60.txt) have the same shape: 5 of 60 outside
# +-0.002 at K = 40 (worst -0.0068), 4 excursions above 1.0 (worst 3.4): a 40-step Adam run on batches of 4 in bf16 storage is that
# noisy on either side.  The gate therefore has two tiers, both over EVERY draw:
#   bf16, hard bounds (no run of either side has come near them since the fix; round 4's seeded run broke all three):
#       |d mIoU| <= 0.015 at K = 40 (worst of 71 HIP runs: 0.0068; round 4: 0.033), <= 0.1 at K = 20 (0.055; 0.256), loss never more than 5.0 off
#       the fp32 trajectory, relative (3.4; 16);
#   bf16, distribution: median within the survey's +-0.002 at K = 40 and within 0.015 at K = 20; at least 9 of the 11 draws within
#       0.003 at K = 40 and inside the emulated reference's K = 20 envelope (the emulated reference itself: 21 of 23 / all);
#   fp32: every draw within 1e-3 (K = 40) / 3e-3 (K = 20), loss trajectory within 0.15.
DRAWS = {"fp32": (0.0, 1e-7, -1e-7, 2e-7, -2e-7, 3e-7, -3e-7),
         "bf16": (0.0, 1e-7, -1e-7, 2e-7, -2e-7, 3e-7, -3e-7, 5e-7, -5e-7, 1e-6, -1e-6)}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_kstep_run_miou_matches_cpu_fp32_reference_run(golden_dir, precision):
    from kurosiwo_amd.snunet import SNUNet_ECAM
    from kurosiwo_amd.trainer import CDTrainStep
    from oracle import metrics_ref, snunet_ref as R
    from oracle.gen_parity_run import BATCH, CHECKPOINTS, HELD_OUT, K_STEPS, TRAIN_TILES, protocol_tiles
    from oracle.seeded import seeded_fill_
    gold = np.load(os.path.join(golden_dir, "snunet_parity_run_ref.npz"))
    env = np.load(os.path.join(golden_dir, "snunet_parity_draws_ref.npz"))
    assert list(gold["protocol"][:4]) == [K_STEPS, TRAIN_TILES, BATCH, HELD_OUT] and CHECKPOINTS == (20, 40)
    # the emulated reference's envelope (deltas against the same fp32 run)
    e20 = env["bf16emu.miou20"] - float(gold["miou20"])
    e40 = env["bf16emu.miou40"] - float(gold["miou40"])
    f40 = env["fp32.miou40"] - float(gold["miou40"])
    e_dev = (np.abs(env["bf16emu.losses"] - gold["losses"]) / gold["losses"]).max(1)
    print(f"imported reference, bf16 storage emulated, {e20.size} draws: K=20 {e20.min():+.5f} ... {e20.max():+.5f} (median {np.median(e20):+.5f}), "
          f"K=40 {e40.min():+.5f} ... {e40.max():+.5f} (median {np.median(e40):+.5f}), loss deviation up to {e_dev.max():.2f}; "
          f"fp32 under the perturbations: K=40 {f40.min():+.5f} ... {f40.max():+.5f}")
    assert e20.size >= 20 and abs(float(env["fp32.miou40"][0]) - float(gold["miou40"])) < 1e-4      # (the fixture is the protocol's)
    dev = torch.device("cuda:0")
    (xA, xB, mask), (eA, eB, emask) = protocol_tiles()

    def evaluate(m):
        m.eval()
        cm = np.zeros((4, 4), np.int64)
        with torch.no_grad():
            for s in range(0, HELD_OUT, 8):
                logits = m(eA[s:s + 8].to(dev), eB[s:s + 8].to(dev)).float().cpu().numpy()
                cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(logits), emask[s:s + 8].numpy())
        m.train()
        return cm, metrics_ref.metrics_from_cm(cm)

    d_miou = {k: [] for k in CHECKPOINTS}
    d_iou = {k: [] for k in CHECKPOINTS}
    runs = []
    model = None
    for pz in DRAWS[precision]:
        sd = seeded_fill_(R.new_state_dict(2, 3, 32))
        if pz:
            sd["conv0_0.conv1.weight"] = sd["conv0_0.conv1.weight"] * (1.0 + pz)
        model = SNUNet_ECAM(2, 3, base_channel=32, precision=precision)
        model.load_state_dict(sd)
        model = model.to(dev).train()
        step = CDTrainStep(model, BATCH, 224, 224, loss_function="ce+dice", class_weights=(1.0, 1.0, 1.0), lr=1e-3)
        losses = []
        for k in range(K_STEPS):
            s = (k % (TRAIN_TILES // BATCH)) * BATCH
            losses.append(float(step.step(xA[s:s + BATCH].to(dev), xB[s:s + BATCH].to(dev), mask[s:s + BATCH].to(dev))[0]))
            if k + 1 in CHECKPOINTS:
                cm, m = evaluate(model)
                g_miou, g_iou, g_cm = float(gold[f"miou{k + 1}"]), gold[f"iou{k + 1}"], gold[f"cm{k + 1}"]
                d_miou[k + 1].append(float(m["miou"]) - g_miou)
                d_iou[k + 1].append(m["iou"][:3] - g_iou[:3])
                print(f"{precision} draw {pz:+g} K={k + 1}: mIoU {m['miou']:.5f} (CPU fp32 {g_miou:.5f}, delta {d_miou[k + 1][-1]:+.5f}); per-class IoU delta "
                      f"{np.array2string(d_iou[k + 1][-1], precision=5)}; pixels in other confusion-matrix cells: {int(np.abs(cm - g_cm).sum()) // 2} of "
                      f"{int(cm.sum())}; loss {losses[-1]:.5f} vs {gold['losses'][k]:.5f}")
        runs.append(np.array(losses))
    runs = np.stack(runs)
    rel = np.abs(runs - gold["losses"]) / gold["losses"]
    for k in CHECKPOINTS:
        print(f"{precision} K={k}: delta mIoU of every draw {np.round(d_miou[k], 5).tolist()}; median {float(np.median(d_miou[k])):+.5f}")
    print(f"{precision}: largest relative deviation of the loss trajectory per draw {np.round(rel.max(1), 3).tolist()}")
    a20, a40 = np.abs(d_miou[20]), np.abs(d_miou[40])
    if precision == "fp32":
        # every draw: the HIP fp32 run IS the reference's run up to the chaos a 1e-7 perturbation shows on the reference itself (5e-4)
        assert a40.max() <= 1e-3, d_miou[40]
        assert a20.max() <= 3e-3, d_miou[20]
        assert np.abs(np.stack(d_iou[40])).max() <= 3e-3 and np.abs(np.stack(d_iou[20])).max() <= 8e-3
        assert rel.max() < 0.15, rel.max(1)
        assert rel[0, 0] < 2e-4                                       # first step of the seeded weights: to rounding
    else:
        # tier 1, every draw: hard bounds
        # (round 6, ADVICE round 5: tightened from 0.015 / 0.1 / 5.0 towards the observed worst cases -- worst of 71 HIP runs 0.0068 / 0.055 /
        # 3.4, the eleven draws here 0.0019 / 0.0114 / 0.47 -- so that a moderate numerical regression fails the hard tier too)
        assert a40.max() <= 8e-3, d_miou[40]
        assert a20.max() <= 6e-2, d_miou[20]
        assert rel.max() <= 3.5, rel.max(1)
        assert np.abs(np.stack(d_iou[40])).max() <= 2.5e-2, d_iou[40]
        # tier 2, the distribution of the draws against the emulated reference's
        assert abs(float(np.median(d_miou[40]))) <= 2e-3, d_miou[40]                       # the survey's gate, on the median
        assert abs(float(np.median(d_miou[20]))) <= 1.5e-2, d_miou[20]
        n = len(d_miou[40])
        assert int((a40 <= 3e-3).sum()) >= n - 2, d_miou[40]                               # emulated reference: 21 of 23
        assert int(((np.array(d_miou[20]) >= e20.min()) & (np.array(d_miou[20]) <= max(e20.max(), 0.002))).sum()) >= n - 2, (d_miou[20], e20.min())
        assert int((rel.max(1) <= 1.0).sum()) >= n - 2, rel.max(1)                         # loss excursions: emulated reference 2 of 23
        assert rel[0, 0] < 2e-2
        # bf16 inference of the trained weights vs fp32 inference of the SAME weights: the eval path is not where bf16 differs
        m32 = SNUNet_ECAM(2, 3, base_channel=32, precision="fp32")
        m32.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        _, a = evaluate(model)
        _, b = evaluate(m32.to(dev))
        print(f"same weights, bf16 vs fp32 inference: mIoU {a['miou']:.5f} vs {b['miou']:.5f}")
        assert abs(float(a["miou"]) - float(b["miou"])) < 3e-4


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_changeformer_kstep_run_matches_cpu_fp32_reference_run(golden_dir, precision):
    """The same K-step protocol for ChangeFormerV6 with the method's shipped optimiser (SGD 6e-4, momentum 0.99, weight decay 1e-5),
    ce+dice on the sigmoid map, stochastic layers at p = 0 on both sides; CPU side = the imported reference
    (oracle/gen_parity_run.py --changeformer -> tests/golden/changeformer_parity_run_ref.npz).  40 SGD steps from the seeded weights
    leave the model early on its learning curve (held-out mIoU 0.24 -> 0.28, mostly background): the gate is the loss trajectory and
    the mIoU / per-class IoU at both checkpoints."""
    from kurosiwo_amd.changeformer import ChangeFormerV6
    from kurosiwo_amd.optim import FusedSGD
    from kurosiwo_amd.trainer import CDTrainStep
    from oracle import changeformer_ref as R, metrics_ref
    from oracle.gen_parity_run import BATCH, CHECKPOINTS, HELD_OUT, K_STEPS, TRAIN_TILES, protocol_tiles
    from oracle.seeded import seeded_fill_
    gold = np.load(os.path.join(golden_dir, "changeformer_parity_run_ref.npz"))
    assert list(gold["protocol"][:4]) == [K_STEPS, TRAIN_TILES, BATCH, HELD_OUT]
    dev = torch.device("cuda:0")
    (xA, xB, mask), (eA, eB, emask) = protocol_tiles()
    model = ChangeFormerV6(2, 3, decoder_softmax=True, embed_dim=256, precision=precision)
    model.drop_rate = model.attn_drop = model.drop_path_rate = 0.0
    model.load_state_dict(seeded_fill_(R.new_state_dict(2, 3, 256)))
    model = model.to(dev).train()
    opt = FusedSGD(model.parameters(), lr=6e-4, momentum=0.99, weight_decay=1e-5)
    step = CDTrainStep(model, BATCH, 224, 224, loss_function="ce+dice", class_weights=(1.0, 1.0, 1.0), optimizer=opt)

    def evaluate(m):
        m.eval()
        cm = np.zeros((4, 4), np.int64)
        with torch.no_grad():
            for s in range(0, HELD_OUT, 8):
                prob = m(eA[s:s + 8].to(dev), eB[s:s + 8].to(dev))[-1].float().cpu().numpy()
                cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(prob), emask[s:s + 8].numpy())
        m.train()
        return cm, metrics_ref.metrics_from_cm(cm)

    # measured on MI355X: fp32 |d mIoU| 1.1e-4, per-class 5e-4, loss trajectory 1e-4; bf16 2.4e-4, 1.1e-3, 4.5e-4 -> the survey's +-0.002 gate holds
    bound_miou, bound_iou, bound_loss = {"fp32": (5e-4, 1.5e-3, 5e-4), "bf16": (2e-3, 3e-3, 2e-3)}[precision]
    losses = []
    for k in range(K_STEPS):
        s = (k % (TRAIN_TILES // BATCH)) * BATCH
        losses.append(float(step.step(xA[s:s + BATCH].to(dev), xB[s:s + BATCH].to(dev), mask[s:s + BATCH].to(dev))[0]))
        if k + 1 in CHECKPOINTS:
            cm, m = evaluate(model)
            g_miou, g_iou = float(gold[f"miou{k + 1}"]), gold[f"iou{k + 1}"]
            d_miou, d_iou = float(m["miou"]) - g_miou, m["iou"][:3] - g_iou[:3]
            print(f"changeformer {precision} K={k + 1}: mIoU {m['miou']:.5f} (CPU fp32 reference {g_miou:.5f}, delta {d_miou:+.5f}); per-class IoU "
                  f"delta {np.array2string(d_iou, precision=5)}; loss {losses[-1]:.5f} vs {gold['losses'][k]:.5f}")
            assert abs(d_miou) <= bound_miou, (k + 1, d_miou)
            assert np.abs(d_iou).max() <= bound_iou, (k + 1, d_iou)
    rel = np.abs(np.array(losses) - gold["losses"]) / gold["losses"]
    print(f"changeformer {precision}: loss trajectory max relative deviation {rel.max():.5f}")
    assert rel.max() < bound_loss, rel


PLATEAU_DRAWS = (0.0, 1e-7, -1e-7, 2e-7, -2e-7, 5e-7, -5e-7, 1e-6, -1e-6)


def test_plateau_protocol_every_bf16_draw_within_the_surveys_gate(golden_dir):
    """the +-0.002 mIoU gate asserted on EVERY draw where the protocol itself is not the noise source.  The
    40-step / batch-4 protocol above is noisy on the reference itself under bf16 storage (3 of 23 emulated draws outside +-0.002); this one
    runs 80 Adam steps on batches of 16 (half the benchmarked per-GPU batch: what an fp32 run of the imported reference fits into the
    build container) and evaluates the 64 held-out tiles after 60 and 80 steps, both far onto the plateau (reference mIoU 0.9892 /
    0.9904, rising 7e-4 per 10 steps).  CPU side = the IMPORTED reference (oracle/gen_parity_run.py --reference-plateau ->
    tests/golden/snunet_parity_plateau_ref.npz; training/change_detection_trainer.py:135-189).  Asserted: every bf16 draw within 0.002
    of the reference's fp32 run at both checkpoints, every fp32 draw within 5e-4."""
    from kurosiwo_amd.snunet import SNUNet_ECAM
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    from kurosiwo_amd.trainer import CDTrainStep
    from oracle import metrics_ref, snunet_ref as R
    from oracle.gen_parity_run import HELD_OUT, HELD_OUT_SEED, P_BATCH, P_CHECKPOINTS, P_K_STEPS, P_TRAIN_TILES, TRAIN_SEED
    from oracle.seeded import seeded_fill_
    gold = np.load(os.path.join(golden_dir, "snunet_parity_plateau_ref.npz"))
    assert list(gold["protocol"][:4]) == [P_K_STEPS, P_TRAIN_TILES, P_BATCH, HELD_OUT] and P_CHECKPOINTS == (60, 80)
    dev = torch.device("cuda:0")
    (xA, xB), mask = cd_inputs(make_batch(P_TRAIN_TILES, seed=TRAIN_SEED), ("pre_event_1", "post_event"))
    (eA, eB), emask = cd_inputs(make_batch(HELD_OUT, seed=HELD_OUT_SEED), ("pre_event_1", "post_event"))

    def evaluate(m):
        m.eval()
        cm = np.zeros((4, 4), np.int64)
        with torch.no_grad():
            for s in range(0, HELD_OUT, 8):
                logits = m(eA[s:s + 8].to(dev), eB[s:s + 8].to(dev)).float().cpu().numpy()
                cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(logits), emask[s:s + 8].numpy())
        m.train()
        return metrics_ref.metrics_from_cm(cm)

    worst = {}
    for precision, draws, bound in (("fp32", PLATEAU_DRAWS[:3], 5e-4), ("bf16", PLATEAU_DRAWS, 2e-3)):
        deltas = {k: [] for k in P_CHECKPOINTS}
        dev_loss = []
        for pz in draws:
            sd = seeded_fill_(R.new_state_dict(2, 3, 32))
            if pz:
                sd["conv0_0.conv1.weight"] = sd["conv0_0.conv1.weight"] * (1.0 + pz)
            model = SNUNet_ECAM(2, 3, base_channel=32, precision=precision)
            model.load_state_dict(sd)
            model = model.to(dev).train()
            step = CDTrainStep(model, P_BATCH, 224, 224, loss_function="ce+dice", class_weights=(1.0, 1.0, 1.0), lr=1e-3)
            losses = []
            for k in range(P_K_STEPS):
                s = (k % (P_TRAIN_TILES // P_BATCH)) * P_BATCH
                losses.append(float(step.step(xA[s:s + P_BATCH].to(dev), xB[s:s + P_BATCH].to(dev), mask[s:s + P_BATCH].to(dev))[0]))
                if k + 1 in P_CHECKPOINTS:
                    m = evaluate(model)
                    deltas[k + 1].append(float(m["miou"]) - float(gold[f"miou{k + 1}"]))
            dev_loss.append(float((np.abs(np.array(losses) - gold["losses"]) / gold["losses"]).max()))
        for k in P_CHECKPOINTS:
            print(f"plateau protocol, {precision}, K={k}: delta mIoU of every draw {np.round(deltas[k], 5).tolist()} (reference {float(gold[f'miou{k}']):.5f})")
        print(f"plateau protocol, {precision}: largest relative deviation of the loss trajectory per draw {np.round(dev_loss, 3).tolist()}")
        worst[precision] = max(max(abs(x) for x in deltas[k]) for k in P_CHECKPOINTS)
        for k in P_CHECKPOINTS:
            assert max(abs(x) for x in deltas[k]) <= bound, (precision, k, deltas[k])
