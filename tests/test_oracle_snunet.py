"""CPU: the SNUNet oracle (oracle/snunet_ref.py) against golden vectors produced by
the real reference model (models/snunet.py imported in oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import snunet_ref as R
from oracle.seeded import seeded_fill_, seeded_labels, seeded_tensor

CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]


def sar_like(name, shape):
    return seeded_tensor(name, shape).clamp_(-2.23, 5.75)


def test_state_dict_inventory_matches_survey():
    # SNUNet-ECAM 12.03 M params, 236 state-dict keys
    spec = R.snunet_state_dict_spec(2, 3, 32)
    assert len(spec) == 236
    n = sum(int(np.prod(s)) for k, s in spec.items() if not R.is_buffer(k))
    assert n == 12_034_819
    spec3 = R.snunet_state_dict_spec(3, 3, 32)
    assert len(spec3) == 236


@pytest.fixture(scope="module")
def small(golden_dir):
    return np.load(os.path.join(golden_dir, "snunet_small.npz"))


@pytest.mark.parametrize("c", [2, 3])
def test_small_eval_and_train_step(small, c):
    tag = f"c{c}"
    B, H, W, bc = 2, 32, 32, 8
    xA = sar_like(f"small.{tag}.xA", (B, c, H, W))
    xB = sar_like(f"small.{tag}.xB", (B, c, H, W))
    lbl = seeded_labels(f"small.{tag}.lbl", (B, H, W))
    sd = seeded_fill_(R.new_state_dict(c, 3, bc))
    with torch.no_grad():
        ev = R.snunet_forward(sd, xA, xB, training=False)
    assert np.abs(ev.numpy() - small[f"{tag}.eval_logits"]).max() < 2e-5

    sd = seeded_fill_(R.new_state_dict(c, 3, bc))
    opt = R.AdamRef(sd, lr=1e-3)
    losses = []
    for step in range(3):
        loss, logits, grads = R.train_step(sd, opt, xA, xB, lbl, CLASS_WEIGHTS, True)
        losses.append(loss)
        if step == 0:
            assert np.abs(logits.numpy() - small[f"{tag}.train_logits"]).max() < 5e-5
            for k, g in grads.items():
                ref = small[f"{tag}.gstat.{k}"]
                assert abs(float(g.double().norm()) - ref[0]) <= 2e-4 * ref[0] + 1e-7, k
                fk = f"{tag}.grad.{k}"
                if fk in small:
                    assert np.abs(g.numpy() - small[fk]).max() <= 2e-4 * np.abs(small[fk]).max() + 1e-8, k
            for k in ("conv0_0.bn1", "conv0_0.bn2", "conv4_0.bn1", "conv0_4.bn2", "conv2_1.bn1"):
                for s in ("running_mean", "running_var"):
                    assert np.abs(sd[f"{k}.{s}"].numpy() - small[f"{tag}.bn.{k}.{s}"]).max() < 1e-5
                assert int(sd[f"{k}.num_batches_tracked"]) == int(small[f"{tag}.bn.{k}.num_batches_tracked"])
            # encoder blocks 0_0..3_0 are called twice per step
            assert int(sd["conv0_0.bn1.num_batches_tracked"]) == 2
            assert int(sd["conv4_0.bn1.num_batches_tracked"]) == 1
            for k in ("conv0_0.conv1.weight", "conv_final.weight", "ca.fc1.weight"):
                assert np.abs(sd[k].numpy() - small[f"{tag}.param1.{k}"]).max() < 2e-6
    assert np.abs(np.array(losses) - small[f"{tag}.losses"]).max() < 5e-4
    sums = np.array([float(sd[k].double().sum()) for k in sd if sd[k].dtype.is_floating_point])
    assert np.abs(sums - small[f"{tag}.param3_sums"]).max() < 5e-2


def test_full_size_eval_logits_and_argmax(golden_dir):
    gold = np.load(os.path.join(golden_dir, "snunet_full.npz"))
    xA = sar_like("full.xA", (1, 2, 224, 224))
    xB = sar_like("full.xB", (1, 2, 224, 224))
    sd = seeded_fill_(R.new_state_dict(2, 3, 32))
    with torch.no_grad():
        logits = R.snunet_forward(sd, xA, xB, training=False)
    scale = float(gold["eval_logits_absmax"])
    assert np.abs(logits[:, :, ::8, ::8].numpy() - gold["eval_logits_sub"]).max() < 1e-4 * scale
    am = logits.argmax(1).numpy().astype(np.uint8)
    margin = gold["eval_margin"].astype(np.float32)
    decisive = margin > 1e-3 * scale
    assert (am[decisive] == gold["eval_argmax"][decisive]).all()
    assert decisive.mean() > 0.99


def test_oracle_kstep_run_equals_the_reference_kstep_run(golden_dir):
    """the 40-step training protocol of the parity gate (oracle/gen_parity_run.py) run on the CPU oracle and on the imported reference
    modules + torch.optim.Adam: the same trajectory (losses to 1e-4 relative, held-out mIoU to 1e-5, a handful of pixels in other
    confusion-matrix cells)"""
    import os
    import numpy as np
    a = np.load(os.path.join(golden_dir, "snunet_parity_run.npz"))
    b = np.load(os.path.join(golden_dir, "snunet_parity_run_ref.npz"))
    assert list(a["protocol"]) == list(b["protocol"])
    assert np.abs(a["losses"] - b["losses"]).max() < 1e-4 * a["losses"].max()
    for k in ("20", "40"):
        assert abs(float(a["miou" + k]) - float(b["miou" + k])) < 1e-5
        assert int(np.abs(a["cm" + k] - b["cm" + k]).sum()) // 2 <= 64
