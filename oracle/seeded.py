"""Deterministic "seeded-fill" weights and inputs shared by the golden
generator, the oracle tests and the GPU parity tests.

Every state-dict tensor is filled from ``torch.Generator(seed=crc32(key)^salt)``
on the CPU, so a test on the GPU box regenerates bit-identical weights without
shipping them.  The distributions are chosen so that
activations stay O(1) through 20+ conv/BN layers and BatchNorm statistics are
non-trivial (mean != 0, var != 1, gamma != 1, beta != 0).
"""
import math
import zlib

import torch


def _gen(key: str, salt: int = 0) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ salt) & 0x7FFFFFFF)
    return g


def seeded_fill_(state_dict, salt: int = 0):
    """In-place deterministic fill of a state dict (any model)."""
    with torch.no_grad():
        for key, t in state_dict.items():
            g = _gen(key, salt)
            if key.endswith("num_batches_tracked"):
                t.zero_()
            elif key.endswith("running_mean"):
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            elif key.endswith("running_var"):
                t.copy_(1.0 + 0.2 * torch.rand(t.shape, generator=g))
            elif key.endswith("pos_embedding") or key.endswith("cls_token"):  # ViT tokens (randn-init in the reference)
                t.copy_(0.5 * torch.randn(t.shape, generator=g))
            elif key.endswith("to_qkv.weight"):  # keep softmax logits O(1): q.k/8 ~ N(0, ~1)
                t.copy_(math.sqrt(0.5 / t.shape[1]) * torch.randn(t.shape, generator=g))
            elif t.dim() >= 2:  # conv / deconv / linear weights
                fan_in = t[0].numel() if t.dim() > 1 else t.numel()
                # ConvTranspose2d weight is [Cin, Cout, kh, kw]: fan-in = Cin*... use dim 0
                if ".up." in key or key.startswith("up."):
                    fan_in = t.shape[0]
                std = math.sqrt(2.0 / max(fan_in, 1))
                t.copy_(std * torch.randn(t.shape, generator=g))
            elif key.endswith("weight"):  # norm gamma
                t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=g))
            else:  # biases, norm beta
                t.copy_(0.05 * torch.randn(t.shape, generator=g))
    return state_dict


def seeded_tensor(name: str, shape, salt: int = 0, kind: str = "randn"):
    g = _gen(name, salt)
    if kind == "randn":
        return torch.randn(shape, generator=g)
    if kind == "rand":
        return torch.rand(shape, generator=g)
    raise ValueError(kind)


def seeded_labels(name: str, shape, salt: int = 0, p_invalid: float = 0.05):
    """int64 labels in {0,1,2,3}; 3 = invalid/ignored (reference ignore_index)."""
    g = _gen(name, salt)
    lbl = torch.randint(0, 3, shape, generator=g, dtype=torch.int64)
    inv = torch.rand(shape, generator=g) < p_invalid
    lbl[inv] = 3
    return lbl
