"""GPU parity of FC-Siam-conc / FC-Siam-diff (row N2): HIP path vs the oracle and the golden vectors of the REAL reference modules
(siam_conc.py / siam_diff.py run with their nn.Dropout2d draws taken from the counter-based stream, oracle/gen_golden.py:gen_fcsiam)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]


def sar_like(name, shape):
    from oracle.seeded import seeded_tensor
    return seeded_tensor(name, shape).clamp_(-2.23, 5.75)


def build(tag, precision):
    from kurosiwo_amd.fcsiam import SiamUnet_conc, SiamUnet_diff
    from oracle import fcsiam_ref as R
    from oracle.seeded import seeded_fill_
    model = (SiamUnet_diff if tag == "diff" else SiamUnet_conc)(2, 3, precision=precision)
    sd = seeded_fill_(R.new_state_dict(2, 3, tag == "diff"))
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    return model.cuda(), sd


def nchw(t, B, h, w):
    return t.float().cpu().reshape(B, h, w, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["conc", "diff"])
def test_eval_and_train_step_vs_reference_golden(golden_dir, tag, precision):
    from oracle import fcsiam_ref as R
    from oracle.seeded import seeded_labels
    from kurosiwo_amd.loss import BCEandDiceLoss
    gold = np.load(os.path.join(golden_dir, f"fcsiam_{tag}.npz"))
    diff = tag == "diff"
    model, sd = build(tag, precision)
    assert list(model.state_dict().keys()) == list(gold["state_dict_keys"])
    S = gold["eval.out"].shape[-1]
    f32 = precision == "fp32"
    # ---- eval
    model.eval()
    with torch.no_grad():
        out = model(sar_like(f"fcsiam.{tag}.eval.x1", (1, 2, S, S)).cuda(), sar_like(f"fcsiam.{tag}.eval.x2", (1, 2, S, S)).cuda())
    # bf16: compare class probabilities (the diff variant returns log-probabilities, whose absolute error is unbounded near p = 0)
    prob = (lambda a: np.exp(a)) if (diff and not f32) else (lambda a: a)
    e = np.abs(prob(out.cpu().numpy()) - prob(gold["eval.out"]))
    assert e.max() < (1e-3 if f32 else 0.15) and e.mean() < (1e-4 if f32 else 2e-2), (e.max(), e.mean())
    # ---- train step with Dropout2d(0.2) ON
    seed, step = (int(v) for v in gold["seed_step"])
    B = 2
    x1 = sar_like(f"fcsiam.{tag}.train.x1", (B, 2, S, S))
    x2 = sar_like(f"fcsiam.{tag}.train.x2", (B, 2, S, S))
    lbl = seeded_labels(f"fcsiam.{tag}.train.lbl", (B, S, S))
    model.train()
    assert model.drop2d == 0.2
    model.manual_seed(seed, step - 1)
    out = model(x1.cuda(), x2.cuda())
    assert model.rng_state().cpu().tolist() == [seed, step]
    plan = model.plan(B, S, S, True, True)
    inter = {}
    with torch.no_grad():
        ref = R.forward(sd, x1, x2, diff, True, {}, (seed, step, 0.2), inter)
    # the plane masks are bit-identical: a dropped (sample, channel) plane is exactly zero on both sides
    for name in ("11_1", "12_2", "43_2", "43d", "22d", "12d"):
        got = nchw(plan.named[name], B, *inter[name].shape[-2:])
        dropped = inter[name].abs().amax((2, 3)) == 0
        assert torch.equal(got.abs().amax((2, 3)) == 0, dropped) or not f32, name
        err = float((got - inter[name]).abs().max() / (inter[name].abs().max() + 1e-12))
        assert err < (5e-4 if f32 else 0.12), (name, err)
    e = np.abs(prob(out.detach().cpu().numpy()) - prob(gold["train.out"]))
    assert e.max() < (1e-3 if f32 else 0.25) and e.mean() < (1e-4 if f32 else 2e-2), (e.max(), e.mean())
    crit = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)
    loss = crit(out, lbl.cuda())
    loss.backward()
    assert abs(float(loss) - float(gold["train.loss"])) < (2e-4 if f32 else 5e-2)
    _, _, ref_grads, stats = R.loss_and_grads(sd, x1, x2, lbl, diff, CLASS_WEIGHTS, stream=(seed, step, 0.2))
    worst, coss = {}, []
    for k, p in model.named_parameters():
        g, r = p.grad.detach().float().cpu(), ref_grads[k]
        if k.startswith("conv") and k.endswith("bias") and k != "conv11d.bias":
            # a conv bias followed directly by BatchNorm has an analytically zero gradient: the reference holds rounding noise
            assert float(g.abs().max()) == 0.0 and float(r.abs().max()) < 1e-3 * float(ref_grads[k[:-4] + "weight"].abs().max()) + 1e-6, k
            continue
        if f32:
            l2 = float((g - r).double().norm() / (r.double().norm() + 1e-30))
            gn = gold[f"gstat.{k}"][0]
            # (diff: |s1 - s2| takes its sign from differences that can be a rounding apart on the two devices)
            if not (l2 < (1e-2 if diff else 3e-3) and abs(float(g.double().norm()) - gn) <= 5e-3 * gn + 1e-7):
                worst[k] = (l2, float(g.double().norm()), gn)
        else:
            cos = float((g.double() * r.double()).sum() / (g.double().norm() * r.double().norm() + 1e-30))
            coss.append(cos)
            if not cos > 0.6:
                worst[k] = cos
    if coss:
        # bf16 storage of every activation + ReLU / max-pool / |a-b| decisions taken on rounded values (the oracle keeps its own fp32
        # decisions): single gradients are noisy at this tile size (2 x 96 x 96 pixels); bound the bulk and the floor
        print(tag, "bf16 gradient cosines: median", float(np.median(coss)), "min", float(np.min(coss)))
        assert float(np.median(coss)) > (0.85 if diff else 0.88), float(np.median(coss))
    assert not worst, f"{tag} {precision}: {len(worst)} params: {dict(list(worst.items())[:10])}"
    msd = model.state_dict()
    for k in ("bn11", "bn43", "bn43d", "bn12d"):
        rt = 1e-3 if f32 else 5e-2
        assert np.abs(msd[f"{k}.running_mean"].cpu().numpy() - gold[f"bn.{k}.running_mean"]).max() < rt * max(1.0, np.abs(gold[f"bn.{k}.running_mean"]).max())
        assert np.abs(msd[f"{k}.running_var"].cpu().numpy() - gold[f"bn.{k}.running_var"]).max() < rt * max(1.0, float(gold[f"bn.{k}.running_var"].max()))
        assert int(msd[f"{k}.num_batches_tracked"]) == int(gold[f"bn.{k}.num_batches_tracked"])


def test_main_entry_siam_conc_end_to_end_tiny(tmp_path, monkeypatch):
    """main.py --method siam-conc on a tiny synthetic set: Adam(1e-5) epoch, checkpoint, reload, test (configs/method/siam-conc)."""
    import shutil
    import main as entry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copytree(os.path.join(root, "configs"), tmp_path / "configs")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("KSMI_SYNTHETIC_TILES", "8,4,4")
    miou = entry.main(["--method", "siam-conc", "--inputs", "pre_event_1", "post_event", "--batch_size", "4"])
    assert 0.0 <= miou <= 100.0
    ck = list((tmp_path / "checkpoints" / "siam-conc").glob("*/best_segmentation.pt"))
    assert ck, "best checkpoint missing"
    assert len(torch.load(ck[0], map_location="cpu")["model_state_dict"]) == 143
