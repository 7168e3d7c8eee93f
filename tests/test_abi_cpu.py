"""CPU: the C-ABI library loads and exports every symbol include/ksmi.h declares; host logic
that needs no GPU (patch selection, chunk tables, model key inventory, loud failure on CPU)."""
import os
import re

import pytest
import torch

from kurosiwo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "ksmi.h")).read()
    declared = set(re.findall(r"\b(ksmi_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ksmi_src", "ksmi_dst"}
    assert declared, "no symbols parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(_lib.SIGNATURES) == declared
    assert lib.ksmi_abi_version() == _lib.ABI_VERSION == 7
    assert lib.ksmi_chunk_elems(_lib.KSMI_BF16) == 32 and lib.ksmi_chunk_elems(_lib.KSMI_F32) == 16


def test_struct_sizes_match_header_layout():
    import ctypes as C
    assert C.sizeof(_lib.Src) == 40 and C.sizeof(_lib.Dst) == 32 and _lib.MAX_CHUNKS == 72
    # chunk tables sit at the tail of the descriptors
    assert _lib.ConvDesc.chunk_src.offset == _lib.ConvDesc.chunk_c0.offset + 2 * _lib.MAX_CHUNKS


def test_choose_patch_covers_image_and_respects_limits():
    from kurosiwo_amd.runtime import choose_patch
    for (H, W) in [(224, 224), (112, 112), (56, 56), (28, 28), (14, 14), (2, 2), (32, 48)]:
        th, tw = choose_patch(H, W)
        assert th * tw <= 256 and (th + 2) * (tw + 2) <= 512
    th, tw = choose_patch(112, 112, 2, 2, 2, 128)
    assert th * tw <= 128 and (2 * th) * (2 * tw) <= 512
    assert choose_patch(224, 224) == (16, 16)


def test_model_state_dict_keys_match_reference_inventory():
    from kurosiwo_amd.snunet import SNUNet_ECAM
    from oracle.snunet_ref import snunet_state_dict_spec
    m = SNUNet_ECAM(2, 3, base_channel=32)
    sd = m.state_dict()
    spec = snunet_state_dict_spec(2, 3, 32)
    assert list(sd.keys()) == list(spec.keys())
    for k, shp in spec.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    assert sum(p.numel() for p in m.parameters()) == 12_034_819
    # init statistics follow snunet.py:110-115
    assert float(sd["conv0_0.bn1.weight"].min()) == 1.0 and float(sd["conv0_0.bn1.bias"].abs().max()) == 0.0
    w = sd["conv0_1.conv1.weight"]
    assert abs(float(w.std()) - (2.0 / (32 * 9)) ** 0.5) < 0.01


def test_no_cpu_fallback():
    from kurosiwo_amd.loss import BCEandDiceLoss
    from kurosiwo_amd.snunet import SNUNet_ECAM
    m = SNUNet_ECAM(2, 3, base_channel=16)
    x = torch.zeros(1, 2, 32, 32)
    with pytest.raises(_lib.KsmiError):
        m(x, x)
    with pytest.raises(_lib.KsmiError):
        BCEandDiceLoss([1, 1, 1], 3, True)(torch.zeros(1, 3, 4, 4), torch.zeros(1, 4, 4, dtype=torch.int64))


def test_load_state_dict_roundtrip_keeps_arena():
    from kurosiwo_amd.snunet import SNUNet_ECAM
    from oracle.seeded import seeded_fill_
    from oracle.snunet_ref import new_state_dict
    m = SNUNet_ECAM(3, 3, base_channel=16)
    sd = seeded_fill_(new_state_dict(3, 3, 16))
    m.load_state_dict(sd)
    assert m._arena_ok()
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
