"""Oracle for BIT-CD (`base_resnet18`, row N2) against the golden vectors of the REAL reference network
(tests/golden/bitcd.npz, oracle/gen_golden.py:gen_bitcd from /root/reference/models/bit_cd.py:define_G)."""
import os

import numpy as np
import torch

from oracle import bitcd_ref as R
from oracle.seeded import seeded_fill_, seeded_labels, seeded_tensor

CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]


def sar_like(name, shape):
    return seeded_tensor(name, shape).clamp_(-2.23, 5.75)


def test_state_dict_eval_and_train_step(golden_dir):
    gold = np.load(os.path.join(golden_dir, "bitcd.npz"))
    spec = R.state_dict_spec(2, 3)
    assert list(spec.keys()) == list(gold["state_dict_keys"])
    assert [",".join(str(d) for d in v) for v in spec.values()] == list(gold["state_dict_shapes"])
    sd = seeded_fill_(R.new_state_dict(2, 3))
    S = gold["eval.out"].shape[-1]
    with torch.no_grad():
        out = R.forward(sd, sar_like("bitcd.eval.x1", (1, 2, S, S)), sar_like("bitcd.eval.x2", (1, 2, S, S)))
    assert np.abs(out.numpy() - gold["eval.out"]).max() < 1e-4 * max(1.0, np.abs(gold["eval.out"]).max())
    x1, x2 = sar_like("bitcd.train.x1", (2, 2, S, S)), sar_like("bitcd.train.x2", (2, 2, S, S))
    lbl = seeded_labels("bitcd.train.lbl", (2, S, S))
    out, loss, grads, stats = R.loss_and_grads(sd, x1, x2, lbl, CLASS_WEIGHTS)
    assert np.abs(out.numpy() - gold["train.out"]).max() < 1e-4 * max(1.0, np.abs(gold["train.out"]).max())
    assert abs(loss - float(gold["train.loss"])) < 1e-5
    for k, g in grads.items():
        ref = gold[f"gstat.{k}"]
        assert abs(float(g.double().norm()) - ref[0]) <= 2e-3 * ref[0] + 1e-6, k
        if f"grad.{k}" in gold:
            assert np.abs(g.numpy() - gold[f"grad.{k}"]).max() <= 2e-3 * np.abs(gold[f"grad.{k}"]).max() + 1e-7, k
    for k in ("resnet.bn1", "resnet.layer2.0.downsample.1", "resnet.layer4.1.bn2", "classifier.1"):
        assert np.abs(stats[f"{k}.running_mean"].numpy() - gold[f"bn.{k}.running_mean"]).max() < 1e-4 * max(1.0, np.abs(gold[f"bn.{k}.running_mean"]).max())
        assert np.abs(stats[f"{k}.running_var"].numpy() - gold[f"bn.{k}.running_var"]).max() < 1e-3 * max(1.0, float(gold[f"bn.{k}.running_var"].max()))
        assert int(stats[f"{k}.num_batches_tracked"]) == int(gold[f"bn.{k}.num_batches_tracked"]) == (1 if k == "classifier.1" else 2)


import pytest


@pytest.mark.parametrize("net_G", list(R.VARIANTS))
def test_base_transformer_oracle_vs_reference_golden(golden_dir, net_G):
    """the three BASE_Transformer variants of define_G against vectors of the REAL reference (oracle/gen_golden.py:gen_bitcd_transformer)"""
    gold = np.load(os.path.join(golden_dir, f"bitcd_{net_G}.npz"))
    spec = R.transformer_state_dict_spec(net_G, 2, 3)
    assert list(spec.keys()) == list(gold["state_dict_keys"])
    assert [",".join(str(d) for d in v) for v in spec.values()] == list(gold["state_dict_shapes"])
    sd = seeded_fill_(R.new_transformer_state_dict(net_G, 2, 3))
    S = gold["eval.out"].shape[-1]
    inter = {}
    with torch.no_grad():
        out = R.transformer_forward(sd, net_G, sar_like("bitcd.eval.x1", (1, 2, S, S)), sar_like("bitcd.eval.x2", (1, 2, S, S)), inter=inter)
    assert np.abs(inter["tokens"].numpy() - gold["eval.tokens"]).max() < 1e-4 * max(1.0, np.abs(gold["eval.tokens"]).max())
    assert np.abs(out.numpy() - gold["eval.out"]).max() < 1e-4 * max(1.0, np.abs(gold["eval.out"]).max())
    x1, x2 = sar_like("bitcd.train.x1", (2, 2, S, S)), sar_like("bitcd.train.x2", (2, 2, S, S))
    lbl = seeded_labels("bitcd.train.lbl", (2, S, S))
    out, loss, grads, stats = R.transformer_loss_and_grads(sd, net_G, x1, x2, lbl, CLASS_WEIGHTS)
    assert np.abs(out.numpy() - gold["train.out"]).max() < 1e-4 * max(1.0, np.abs(gold["train.out"]).max())
    assert abs(loss - float(gold["train.loss"])) < 1e-5
    full = 0
    for k, g in grads.items():
        ref = gold[f"gstat.{k}"]
        assert abs(float(g.double().norm()) - ref[0]) <= 2e-3 * ref[0] + 1e-6, k
        if f"grad.{k}" in gold:
            assert np.abs(g.numpy() - gold[f"grad.{k}"]).max() <= 2e-3 * np.abs(gold[f"grad.{k}"]).max() + 1e-7, k
            full += 1
    assert full >= 20
    for k in ("resnet.layer4.0.conv1.weight", "resnet.fc.weight"):                  # cut off by resnet_stages_num = 4: no gradient
        assert float(grads[k].abs().max()) == 0.0 and gold[f"gstat.{k}"][0] == 0.0
    for k in ("resnet.bn1", "resnet.layer3.1.bn2", "resnet.layer4.1.bn2", "classifier.1"):
        assert np.abs(stats.get(f"{k}.running_mean", sd[f"{k}.running_mean"]).numpy() - gold[f"bn.{k}.running_mean"]).max() < 1e-4 * max(1.0, np.abs(gold[f"bn.{k}.running_mean"]).max())
        want = 0 if "layer4" in k else (1 if k == "classifier.1" else 2)
        assert int(stats.get(f"{k}.num_batches_tracked", sd[f"{k}.num_batches_tracked"])) == int(gold[f"bn.{k}.num_batches_tracked"]) == want
