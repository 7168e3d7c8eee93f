"""CPU: the loss oracle (oracle/loss_ref.py, closed-form float64 numpy) against the
golden vectors produced by the real reference (oracle/gen_golden.py), and the
torch-autograd restatement used inside the oracle's train step."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref
from oracle.seeded import seeded_labels, seeded_tensor
from oracle.snunet_ref import torch_ce_dice

CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]


def _cases():
    kat_logits = torch.tensor([[[[1, -.5], [.25, 2]], [[0, .5], [-1, .5]], [[-1, 1.5], [.75, -2]]]], dtype=torch.float32)
    kat_lbl = torch.tensor([[[0, 2], [3, 1]]], dtype=torch.int64)
    return {
        "kat": (kat_logits, kat_lbl),
        "rand": (seeded_tensor("loss.rand.logits", (2, 3, 16, 16)) * 2.0, seeded_labels("loss.rand.labels", (2, 16, 16))),
        "big": (seeded_tensor("loss.big.logits", (3, 3, 64, 48)) * 4.0, seeded_labels("loss.big.labels", (3, 64, 48), p_invalid=0.3)),
    }


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "loss_cases.npz"))


def test_kat_loss_1_survey_values(gold):
    # SURVEY.md §4 KAT-loss-1 (measured on the reference)
    assert abs(float(gold["kat.unit.dice"]) - 0.5372734070) < 1e-7
    assert abs(float(gold["kat.unit.ce"]) - 0.8438295722) < 1e-7
    assert abs(float(gold["kat.unit.total"]) - 1.3811030388) < 1e-7
    assert abs(float(gold["kat.cw.total"]) - 1.7566509247) < 1e-7


@pytest.mark.parametrize("case", ["kat", "rand", "big"])
@pytest.mark.parametrize("wname", ["unit", "cw"])
def test_numpy_oracle_matches_reference(gold, case, wname):
    x, t = _cases()[case]
    w = [1.0, 1.0, 1.0] if wname == "unit" else CLASS_WEIGHTS
    r = loss_ref.ce_dice_forward(x.numpy(), t.numpy(), w, with_grad=True)
    for k in ("dice", "ce", "total"):
        assert abs(r[k] - float(gold[f"{case}.{wname}.{k}"])) < 2e-6 * max(1.0, abs(r[k])), k
    g = gold[f"{case}.{wname}.grad"]
    assert np.abs(r["grad"] - g).max() < 2e-6 * max(1e-3, np.abs(g).max()) + 1e-9
    r2 = loss_ref.ce_dice_forward(x.numpy(), t.numpy(), w, with_dice=False, with_grad=True)
    assert abs(r2["total"] - float(gold[f"{case}.{wname}.ce_only"])) < 2e-6
    g2 = gold[f"{case}.{wname}.ce_only_grad"]
    assert np.abs(r2["grad"] - g2).max() < 2e-6 * max(1e-3, np.abs(g2).max()) + 1e-9


@pytest.mark.parametrize("case", ["kat", "rand"])
def test_torch_restatement_matches_reference(gold, case):
    x, t = _cases()[case]
    xx = x.clone().requires_grad_(True)
    loss = torch_ce_dice(xx, t, CLASS_WEIGHTS, True)
    loss.backward()
    assert abs(float(loss) - float(gold[f"{case}.cw.total"])) < 1e-6
    assert np.abs(xx.grad.numpy() - gold[f"{case}.cw.grad"]).max() < 1e-7


def test_ignored_pixel_still_gets_dice_gradient(gold):
    # SURVEY.md §4 note: label 3 is relabelled class 0 for the dice term
    g = gold["kat.unit.grad"]
    assert np.abs(g[0, :, 1, 0]).max() > 1e-3


def test_all_ignored_is_nan_like_reference():
    x = np.zeros((1, 3, 2, 2), np.float32)
    t = np.full((1, 2, 2), 3, np.int64)
    with np.errstate(all="ignore"):
        r = loss_ref.ce_dice_forward(x, t)
    assert np.isnan(r["ce"])  # nn.CrossEntropyLoss: 0/0 when every target is ignored


def test_loss_from_config_dispatch():
    x, t = _cases()["rand"]
    tr = loss_ref.loss_from_config({"loss_function": "cross_entropy", "class_weights": CLASS_WEIGHTS}, "train")(x.numpy(), t.numpy())
    va = loss_ref.loss_from_config({"loss_function": "cross_entropy", "class_weights": CLASS_WEIGHTS}, "val")(x.numpy(), t.numpy())
    assert abs(tr["total"] - va["total"]) > 1e-3     # train weighted, val unweighted (utilities.py:314-321)
    with pytest.raises(NotImplementedError):
        loss_ref.loss_from_config({"loss_function": "focal"})
