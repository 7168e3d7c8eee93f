"""Host-side geometry of the 3x3 weight gradient (csrc/wgrad3.hip: ksmi_wgrad3_geom behind ksmi_conv_wgrad_workspace): split counts
at the SNUNet bs = 32 shapes (models/snunet.py:15-29 backward), and the invariants the parity tests rely on -- the ring depth never
changes the slab layout, a forced workgroup count is honoured.  No GPU: descriptors are built over CPU tensors and never launched."""
import os

import pytest
import torch

from kurosiwo_amd.runtime import SrcSpec, make_wgrad


def _nsplit(B, H, cs, N, **env):
    from kurosiwo_amd import _lib
    for k, v in env.items():
        _lib.set_knob(k, v)                                       # (run-time knobs of the launcher: include/ksmi.h ksmi_set_knob)
    try:
        dt = torch.bfloat16
        xs = [torch.empty((1,), dtype=dt) for _ in cs]          # (pointers only: the geometry reads shapes from the descriptor)
        srcs = [SrcSpec(x, c) for x, c in zip(xs, cs)]
        for s, c in zip(srcs, cs):
            s.C = c
        dy = torch.empty((1,), dtype=dt)
        K = sum(cs)
        grad = torch.empty((1,), dtype=torch.float32)
        d, ws = make_wgrad(srcs, dy, N, 0, N, grad, 9, K * 9, 1, 0, B, H, H, H, H, 3, 3, 1, 1, dt)
        npad = (N + 15) // 16 * 16
        assert ws % (9 * K * npad * 4) == 0
        return d.nsplit
    finally:
        for k in env:
            _lib.set_knob(k, None)


def _tiles(K, N):
    wc = 2 if K == 32 else 4
    npad = (N + 15) // 16 * 16
    ntl = 64 if npad >= 48 else 32
    return -(-(K // 32) // (wc // 2)) * -(-npad // ntl)


SHAPES = [(224, [32], 32), (224, [32, 32, 64], 32), (112, [64], 64), (112, [64, 64, 128], 64), (56, [128], 128),
          (56, [128, 128, 256], 128), (28, [256], 256), (28, [256, 256, 512], 256), (14, [512], 512)]


@pytest.mark.parametrize("H,cs,N", SHAPES)
def test_split_count_is_independent_of_the_ring_depth(H, cs, N):
    base = _nsplit(32, H, cs, N)
    assert base >= 1
    for nst in (2, 3, 4):
        assert _nsplit(32, H, cs, N, KSMI_WGRAD3_NST=nst) == base


@pytest.mark.parametrize("H,cs,N", SHAPES)
def test_forced_workgroup_count(H, cs, N):
    K = sum(cs)
    tiles = _tiles(K, N)
    for wgs in (tiles, 4 * tiles, 512):
        ns = _nsplit(32, H, cs, N, KSMI_WGRAD3_WGS=wgs)
        want = max(1, wgs // tiles)
        assert 1 <= ns <= want                                  # (whole patches per split: never more splits than asked for)
        assert ns * tiles <= max(wgs, tiles)


def test_square_layers_of_levels_2_to_4_take_two_workgroups_per_cu():
    """profiles/r04_wgrad3_wgs.txt: K = N = 128 / 256 / 512 run 10-14 % faster on 512 workgroups than on 256 (one wave per SIMD exposes
    every barrier and DMA wait); the cost model's one-per-CU efficiency was refitted to reproduce that"""
    for H, C in ((56, 128), (28, 256), (14, 512)):
        ns = _nsplit(32, H, [C], C)
        assert 256 < ns * _tiles(C, C) <= 512, (H, C, ns)


@pytest.mark.parametrize("B,H,W,C", [(32, 112, 112, 64), (32, 56, 56, 128), (32, 28, 28, 256), (32, 14, 14, 512), (4, 24, 20, 64), (2, 7, 9, 512)])
def test_up_weight_gradient_workspace_is_whole_slabs(B, H, W, C):
    """ksmi_up_wgrad_workspace (csrc/gemm2.hip: up_wgrad_geom): whole [C][4C] fp32 slabs, at least 8 reduction steps of 64 rows per
    split, ~two workgroups per CU at the full-size shapes (models/snunet.py:32-46 backward)"""
    from kurosiwo_amd import _lib
    lib = _lib.load()
    assert lib.ksmi_up_wgrad_supported(B, H, W, C, 1) == 1
    ws = lib.ksmi_up_wgrad_workspace(B, H, W, C)
    slab = C * 4 * C * 4
    assert ws % slab == 0
    nsplit = ws // slab
    rows = B * H * W
    steps = -(-rows // 64)
    assert 1 <= nsplit <= max(1, steps // 8)
    tiles = (4 * C // 128) * (C // 64)
    assert nsplit * tiles <= 512 + tiles
    if rows >= 32 * 14 * 14 and steps // 8 >= 512 // tiles:
        assert nsplit * tiles >= 256
