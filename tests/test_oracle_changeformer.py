"""CPU: the ChangeFormerV6 oracle (oracle/changeformer_ref.py) against golden vectors produced by the real reference
(models/changeformer.py imported in oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import changeformer_ref as R
from oracle.seeded import seeded_fill_, seeded_labels, seeded_tensor

CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]


def sar_like(name, shape):
    return seeded_tensor(name, shape).clamp_(-2.23, 5.75)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "changeformer.npz"))


def test_state_dict_inventory(gold):
    # 373 state-dict keys, 41 035 255 parameters (c = 2)
    spec = R.changeformer_state_dict_spec(2, 3, 256)
    assert list(spec.keys()) == list(gold["state_dict_keys"])
    assert [",".join(str(d) for d in s) for s in spec.values()] == list(gold["state_dict_shapes"])
    assert len(spec) == 373
    assert sum(int(np.prod(s)) for k, s in spec.items() if not R.is_buffer(k)) == 41_035_255


def test_eval_forward(gold):
    torch.set_num_threads(min(16, torch.get_num_threads()))
    sd = seeded_fill_(R.new_state_dict(2, 3, 256))
    x1 = sar_like("changeformer.eval.x1", (1, 2, 224, 224))
    x2 = sar_like("changeformer.eval.x2", (1, 2, 224, 224))
    inter = {}
    with torch.no_grad():
        outs = R.changeformer_forward(sd, x1, x2, training=False, inter=inter)
    assert [tuple(o.shape) for o in outs] == [(1, 3, 7, 7), (1, 3, 14, 14), (1, 3, 28, 28), (1, 3, 56, 56), (1, 3, 224, 224)]
    for i in range(4):
        assert np.abs(outs[i].numpy() - gold[f"eval.out{i}"]).max() < 1e-4
        assert np.abs(inter[f"A.f{i + 1}"][:, ::4, ::2, ::2].numpy() - gold[f"eval.feat{i + 1}_sub"]).max() < 2e-3
    assert np.abs(outs[4][:, :, ::8, ::8].numpy() - gold["eval.out4_sub"]).max() < 1e-4
    confident = gold["eval.margin"].astype(np.float32) > 1e-3
    assert (outs[4].argmax(1).numpy().astype(np.uint8) == gold["eval.argmax"])[confident].all()


def test_train_step(gold):
    torch.set_num_threads(min(16, torch.get_num_threads()))
    sd = seeded_fill_(R.new_state_dict(2, 3, 256))
    x1 = sar_like("changeformer.train.x1", (2, 2, 224, 224))
    x2 = sar_like("changeformer.train.x2", (2, 2, 224, 224))
    lbl = seeded_labels("changeformer.train.lbl", (2, 224, 224))
    outs, loss, grads, new_stats = R.loss_and_grads(sd, x1, x2, lbl, CLASS_WEIGHTS, True)
    for i in range(4):
        assert np.abs(outs[i].numpy() - gold[f"train.out{i}"]).max() < 1e-4
    assert np.abs(outs[4][:, :, ::8, ::8].numpy() - gold["train.out4_sub"]).max() < 1e-4
    assert abs(loss - float(gold["train.loss"])) < 1e-5
    for k, g in grads.items():
        ref = gold[f"gstat.{k}"]
        assert abs(float(g.double().norm()) - ref[0]) <= 2e-3 * ref[0] + 1e-7, k
        fk = f"grad.{k}"
        if fk in gold:
            assert np.abs(g.numpy() - gold[fk]).max() <= 2e-3 * np.abs(gold[fk]).max() + 1e-8, k
    for k in ("TDec_x2.diff_c4.2", "TDec_x2.diff_c1.2", "TDec_x2.make_pred_c2.2", "TDec_x2.linear_fuse.1"):
        assert np.abs(new_stats[f"{k}.running_mean"].numpy() - gold[f"bn.{k}.running_mean"]).max() < 1e-4
        assert np.abs(new_stats[f"{k}.running_var"].numpy() - gold[f"bn.{k}.running_var"]).max() < 1e-3 * max(1.0, float(gold[f"bn.{k}.running_var"].max()))
        assert int(new_stats[f"{k}.num_batches_tracked"]) == int(gold[f"bn.{k}.num_batches_tracked"])


def test_stochastic_layers_vs_reference_golden(golden_dir):
    """Dropout / attention dropout / DropPath at the reference's probabilities (changeformer.py:651-653): the oracle with the masks of
    oracle/rng_ref.py against the reference's own modules run with the same counter-based draws (gen_golden.py:gen_changeformer_drop)."""
    from oracle import rng_ref as G
    gold = np.load(os.path.join(golden_dir, "changeformer_drop.npz"))
    seed, step = (int(v) for v in gold["seed_step"])
    torch.set_num_threads(min(16, torch.get_num_threads()))
    sd = seeded_fill_(R.new_state_dict(2, 3, 256))
    x1 = sar_like("changeformer.drop.x1", (2, 2, 224, 224))
    x2 = sar_like("changeformer.drop.x2", (2, 2, 224, 224))
    lbl = seeded_labels("changeformer.drop.lbl", (2, 224, 224))
    outs, loss, grads, _ = R.loss_and_grads(sd, x1, x2, lbl, CLASS_WEIGHTS, True, stream=G.DropStream(seed, step))
    for i in range(4):
        assert np.abs(outs[i].numpy() - gold[f"train.out{i}"]).max() < 1e-4
    assert np.abs(outs[4][:, :, ::8, ::8].numpy() - gold["train.out4_sub"]).max() < 1e-4
    assert abs(loss - float(gold["train.loss"])) < 1e-5
    for k, g in grads.items():
        ref = gold[f"gstat.{k}"]
        assert abs(float(g.double().norm()) - ref[0]) <= 2e-3 * ref[0] + 1e-7, k
        fk = f"grad.{k}"
        if fk in gold:
            assert np.abs(g.numpy() - gold[fk]).max() <= 2e-3 * np.abs(gold[fk]).max() + 1e-8, k
    # the masks matter: without them the loss is a different number
    # (the seeded weights make the loss itself insensitive to the encoder; its gradients are not)
    _, loss0, grads0, _ = R.loss_and_grads(sd, x1, x2, lbl, CLASS_WEIGHTS, True)
    k = "Tenc_x2.block2.1.mlp.fc2.weight"
    rel = float((grads0[k] - grads[k]).norm() / grads[k].norm())
    print("loss", loss, "without masks", loss0, "relative change of", k, rel)
    assert rel > 0.05


def test_rng_stream_statistics():
    """keep rate and independence of the counter-based draws (oracle/rng_ref.py = csrc/common.h ksmi_rng_*)"""
    from oracle import rng_ref as G
    m = G.scale_mask(7, 3, 8 * 5 + G.SITE_MLP1, 0.1, 0, (1 << 20,))
    keep = (m > 0).mean()
    assert abs(keep - 0.9) < 2e-3 and np.allclose(m[m > 0], 1 / 0.9)
    m2 = G.scale_mask(7, 4, 8 * 5 + G.SITE_MLP1, 0.1, 0, (1 << 20,))          # next step: a fresh mask
    m3 = G.scale_mask(7, 3, 8 * 5 + G.SITE_MLP2, 0.1, 0, (1 << 20,))          # another site: a fresh mask
    for other in (m2, m3):
        both = ((m > 0) & (other > 0)).mean()
        assert abs(both - 0.81) < 3e-3
    # windows of one stream agree with the whole (element index = position in the batched tensor)
    w = G.scale_mask(7, 3, 8 * 5 + G.SITE_MLP1, 0.1, 1000, (5000,))
    assert np.array_equal(w, m[1000:6000])
    # lag-1 independence
    k = (m > 0).astype(np.float64)
    assert abs(np.corrcoef(k[:-1], k[1:])[0, 1]) < 5e-3


def test_slc_four_band_inputs(golden_dir):
    """BASELINE.json configs[3] as written (SLC, 4 bands per date -> input_nc = 4, utilities/utilities.py:386-390): the oracle against the
    reference's golden vectors for that configuration (eval outputs, train loss, gradient norms, the first patch-embedding gradient)."""
    torch.set_num_threads(min(16, torch.get_num_threads()))
    gold = np.load(os.path.join(golden_dir, "changeformer_slc.npz"))
    spec = R.changeformer_state_dict_spec(4, 3, 256)
    assert list(spec.keys()) == list(gold["state_dict_keys"])
    assert [",".join(str(d) for d in s) for s in spec.values()] == list(gold["state_dict_shapes"])
    sd = seeded_fill_(R.new_state_dict(4, 3, 256))
    x1 = sar_like("changeformer.slc.eval.x1", (1, 4, 224, 224))
    x2 = sar_like("changeformer.slc.eval.x2", (1, 4, 224, 224))
    with torch.no_grad():
        outs = R.changeformer_forward(sd, x1, x2, training=False)
    for i in range(4):
        assert np.abs(outs[i].numpy() - gold[f"eval.out{i}"]).max() < 1e-4
    assert np.abs(outs[4][:, :, ::8, ::8].numpy() - gold["eval.out4_sub"]).max() < 1e-4
    x1 = sar_like("changeformer.slc.train.x1", (2, 4, 224, 224))
    x2 = sar_like("changeformer.slc.train.x2", (2, 4, 224, 224))
    lbl = seeded_labels("changeformer.slc.train.lbl", (2, 224, 224))
    outs, loss, grads, _ = R.loss_and_grads(sd, x1, x2, lbl, CLASS_WEIGHTS, True)
    assert abs(loss - float(gold["train.loss"])) < 1e-5
    for k, g in grads.items():
        ref = gold[f"gstat.{k}"]
        assert abs(float(g.double().norm()) - ref[0]) <= 2e-3 * ref[0] + 1e-7, k
    k = "Tenc_x2.patch_embed1.proj.weight"
    assert tuple(grads[k].shape) == (64, 4, 7, 7)
    assert np.abs(grads[k].numpy() - gold[f"grad.{k}"]).max() <= 2e-3 * np.abs(gold[f"grad.{k}"]).max() + 1e-8
