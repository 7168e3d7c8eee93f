"""CPU: the FloodViT oracle (oracle/vit_ref.py) against golden vectors produced by the real reference
(models/vision_transformer.py ViT wrapped by models/model_utilities.py FinetunerSegmentation + Decoder,
imported in oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import vit_ref as V
from oracle.seeded import seeded_fill_, seeded_labels, seeded_tensor

CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]
SMALL = dict(channels=6, image_size=224, patch_size=16, dim=1024, depth=2, heads=4, mlp_dim=512)
FULL = dict(channels=6, image_size=224, patch_size=16, dim=1024, depth=24, heads=16, mlp_dim=2048)


def sar_like(name, shape):
    return seeded_tensor(name, shape).clamp_(-2.23, 5.75)


def test_state_dict_inventory():
    # encoder 276 keys incl. mlp_head (2) -> 274 + 6 head keys under FinetunerSegmentation
    spec = V.floodvit_state_dict_spec(**FULL)
    assert len(spec) == 280
    n = sum(int(np.prod(s)) for s in spec.values())
    enc = sum(int(np.prod(s)) for k, s in spec.items() if k.startswith("model."))
    assert enc + 1024 * 1000 + 1000 == 204_332_008            # 204.3 M encoder parameters incl. mlp_head
    assert n == enc + 1024 * 128 * 16 + 128 + 128 * 64 * 16 + 64 + 64 * 3 * 16 + 3


@pytest.mark.parametrize("tag,hp,B,head", [("small", SMALL, 2, "decoder"), ("full", FULL, 1, "decoder"),
                                           ("small_mlp", SMALL, 1, "mlp"), ("small_linear", SMALL, 1, "linear")])
def test_oracle_matches_reference_golden(golden_dir, tag, hp, B, head):
    """head = "mlp" / "linear": FinetunerSegmentation's other two heads (model_utilities.py:59-72; bilinear to 224^2, then 1x1 convs)"""
    torch.set_num_threads(min(8, torch.get_num_threads()))      # many small ops: more threads only add sync overhead
    gold = np.load(os.path.join(golden_dir, f"floodvit_{tag}.npz"))
    spec = V.floodvit_state_dict_spec(**hp, head=head)
    assert list(gold["state_dict_keys"]) == list(spec.keys())
    sd = seeded_fill_(V.new_state_dict(**hp, head=head))
    x = sar_like(f"floodvit.{tag}.x", (B, hp["channels"], 224, 224))
    lbl = seeded_labels(f"floodvit.{tag}.lbl", (B, 224, 224))
    with torch.no_grad():
        tok = V.floodvit_forward(sd, x, hp["heads"], return_tokens=True)
    assert np.abs(tok[:, ::7, ::16].numpy() - gold["tokens_sub"]).max() < 2e-4
    logits, loss, grads = V.loss_and_grads(sd, x, lbl, hp["heads"], CLASS_WEIGHTS)
    assert np.abs(logits[:, :, ::8, ::8].numpy() - gold["logits_sub"]).max() < 5e-4
    assert abs(loss - float(gold["loss"])) < 1e-5
    am = logits.argmax(1).numpy().astype(np.uint8)
    confident = gold["margin"].astype(np.float32) > 1e-3
    assert (am == gold["argmax"])[confident].all()
    for k, g in grads.items():
        ref = gold[f"gstat.{k}"]
        assert abs(float(g.double().norm()) - ref[0]) <= 1e-3 * ref[0] + 1e-7, k
        fk = f"grad.{k}"
        if fk in gold:
            assert np.abs(g.numpy() - gold[fk]).max() <= 1e-3 * np.abs(gold[fk]).max() + 1e-8, k
