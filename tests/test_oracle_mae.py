"""CPU: the MAE oracle (oracle/mae_ref.py) against the golden vectors generated from the reference's models/mae.py
(oracle/gen_golden.py gen_mae; decoder class = the reference's in-tree Transformer), and the host-side MAE module's
state-dict surface."""
import os

import numpy as np
import pytest
import torch

from oracle import mae_ref
from oracle.seeded import seeded_fill_, seeded_tensor

# (the same values as oracle/gen_golden.py MAE_SMALL / sar_like; that module imports /root/reference and must not be imported here)
MAE_SMALL = dict(channels=2, image_size=224, patch_size=16, dim=1024, depth=2, heads=4, mlp_dim=512, decoder_dim=512, decoder_depth=2,
                 decoder_heads=4)


def sar_like(name, shape):
    return seeded_tensor(name, shape).clamp_(-2.23, 5.75)


GOLD = os.path.join(os.path.dirname(__file__), "golden", "mae_small.npz")


def _module():
    from kurosiwo_amd.floodvit import ViT
    from kurosiwo_amd.mae import MAE
    hp = MAE_SMALL
    enc = ViT(image_size=hp["image_size"], patch_size=hp["patch_size"], num_classes=1000, dim=hp["dim"], depth=hp["depth"], heads=hp["heads"],
              mlp_dim=hp["mlp_dim"], channels=hp["channels"])
    return MAE(encoder=enc, masking_ratio=0.75, decoder_dim=hp["decoder_dim"], decoder_depth=hp["decoder_depth"], decoder_heads=hp["decoder_heads"],
               precision="fp32")


def test_state_dict_keys_match_reference():
    g = np.load(GOLD)
    m = _module()
    assert list(m.state_dict().keys()) == [str(k) for k in g["state_dict_keys"]]
    sd = m.state_dict()
    m.load_state_dict(sd)                                      # aliases are accepted and ignored
    assert sd["patch_to_emb.1.weight"].data_ptr() == sd["encoder.to_patch_embedding.2.weight"].data_ptr()


def test_oracle_matches_reference_golden():
    g = np.load(GOLD)
    hp = MAE_SMALL
    m = _module()
    seeded_fill_(m.state_dict())
    sd = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith("patch_to_emb.")}
    x = sar_like("mae.small.x", (2, hp["channels"], 224, 224))
    idx = torch.from_numpy(g["rand_indices"])
    loss, grads, _ = mae_ref.loss_and_grads(sd, x, idx, hp["heads"], hp["decoder_heads"])
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for key in g.files:
        if key.startswith("grad."):
            ref = torch.from_numpy(g[key])
            got = grads[key[5:]]
            assert torch.allclose(got, ref, rtol=2e-4, atol=1e-6 + 1e-4 * float(ref.abs().max())), key
        elif key.startswith("gstat."):
            k = key[6:]
            if k not in grads:
                continue
            gd = grads[k].double()
            ref = g[key]
            assert abs(float(gd.norm()) - ref[0]) <= 2e-4 * ref[0] + 1e-7, key
            assert abs(float(gd.abs().max()) - ref[2]) <= 2e-4 * ref[2] + 1e-7, key
    # parameters MAE.forward never touches get no gradient (mae.py:54-124)
    for k in ("encoder.cls_token", "encoder.mlp_head.weight", "encoder.mlp_head.bias"):
        assert float(grads[k].abs().max()) == 0.0
