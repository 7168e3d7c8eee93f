"""CPU: inventory of the Unet(resnet18) restatement (row U1; PARITY UNPINNED -- see oracle/unet_ref.py)."""
import numpy as np
import torch

from oracle import unet_ref as U
from oracle.seeded import seeded_fill_, seeded_tensor


def test_inventory_and_shapes():
    spec = U.unet_state_dict_spec(3, 3)
    # segmentation_models_pytorch documents Unet(resnet18) at 14.3 M parameters (3 input channels, 1 class: 14 328 209; +2 classes of
    # the 16-channel 3x3 head = 290 more)
    n = sum(int(np.prod(s)) for k, s in spec.items() if not U.is_buffer(k))
    assert n == 14_328_209 + 2 * (16 * 9 + 1)
    assert len(U.unet_state_dict_spec(2, 3)) == 182
    sd = seeded_fill_(U.new_state_dict(2, 3))
    x = seeded_tensor("unet.cpu.x", (1, 2, 64, 64))
    with torch.no_grad():
        y = U.unet_forward(sd, x)
    assert y.shape == (1, 3, 64, 64) and torch.isfinite(y).all()


def test_module_keys_match_restatement():
    from kurosiwo_amd.unet import Unet
    m = Unet("resnet18", encoder_weights=None, in_channels=2, classes=3)
    assert list(m.state_dict().keys()) == list(U.unet_state_dict_spec(2, 3).keys())
