"""Host-side dispatch of the convolution descriptors (no GPU: descriptors over CPU tensors, never launched).
ksmi_conv_stats_rows() answers with the statistics-row count of the kernel that WILL run a descriptor: one row per workgroup of a
persistent kernel (igemm3 / igemm4: at most a few hundred), one per tile on the tile kernels (igemm2 and the first-generation kernel:
B x tiles = thousands at 224 x 224).  Round 5 found whole model families silently on the round-1 kernels because of descriptor
properties no persistent kernel accepted (a statistics buffer sized for the tile kernel, 16-channel sources, 2- / 3-class heads); this
file pins the descriptors of those layers to the persistent kernels, and ksmi_conv_wgrad_fuses_bias to its two modes."""
import ctypes as C

import pytest
import torch

from kurosiwo_amd import _lib
from kurosiwo_amd.runtime import DT, SrcSpec, conv_grid_m, conv_npad, conv_stats_rows, make_conv, make_wgrad

BF = torch.bfloat16


def _desc(B, H, W, cs, N, out_c=None, k_real=None, mask=False, ndst=1, KH=3):
    xs = [torch.empty((16,), dtype=BF) for _ in cs]
    srcs = [SrcSpec(x, c, k_real=(k_real if k_real is not None else None)) for x, c in zip(xs, cs)]
    for s, c in zip(srcs, cs):
        s.C = c
    oc = out_c or N
    out = torch.empty((16,), dtype=BF)
    if ndst == 1:
        dsts = [(out, oc, 0, 0, N, 0)]
    else:
        per = N // ndst
        dsts = [(torch.empty((16,), dtype=BF), per, 0, i * per, per, 0) for i in range(ndst)]
    mk = None
    if mask:
        m = torch.empty((16,), dtype=BF)
        f = torch.empty((N,), dtype=torch.float32)
        mk = (m, f, f, f, f)
    d, table = make_conv(srcs, dsts, out, None, None, B, H, W, H, W, KH, KH, 1, KH // 2 if KH == 3 else 0, N, BF, mask=mk)
    d._keep = (xs, out, dsts, mk)
    return d


def _persistent(d):
    """True when a persistent kernel (one statistics row per workgroup) will run the descriptor"""
    tiles = conv_grid_m(d)
    rows = conv_stats_rows(d, BF)
    assert 1 <= rows <= tiles
    return rows < tiles


@pytest.mark.parametrize("name,kw", [
    ("SNUNet level 0, 32 -> 32", dict(cs=[32], N=32)),
    ("SNUNet level 0 dense skip, 128 -> 32", dict(cs=[32, 32, 64], N=32)),
    ("Unet decoder block 5 conv2 / FC-Siam conv12, 16 -> 16", dict(cs=[16], N=16)),
    ("FC-Siam conv21 input gradient, 16 -> 32", dict(cs=[16], N=32)),
    ("FC-Siam conv11 on the 8-channel image copy (2 real channels)", dict(cs=[8], N=16, k_real=2)),
    ("Unet segmentation head, 16 -> 3 into an 8-channel-stride tensor", dict(cs=[16], N=3, out_c=8)),
    ("head input gradient, 3 (stride 8) -> 16", dict(cs=[8], N=16, k_real=3)),
    ("ChangeFormer change_probability, 256 -> 2", dict(cs=[256], N=2, out_c=8)),
    ("Unet decoder block 5 conv2 input gradient with the mask epilogue, 16 -> 16", dict(cs=[16], N=16, mask=True)),
    ("FC-Siam conv12d input gradient into three 16-channel tensors", dict(cs=[16], N=48, ndst=3)),
    ("FC-Siam-diff conv12d, 16 + 16 -> 16", dict(cs=[16, 16], N=16)),
])
def test_full_resolution_layers_run_on_a_persistent_kernel(name, kw):
    lib = _lib.load()
    d = _desc(32, 224, 224, **kw)
    assert _persistent(d), name
    if kw["N"] < 16:
        assert d.Npad == 32 == conv_npad(kw["N"])


def test_descriptors_the_persistent_kernels_must_refuse():
    # three half-empty chunks: more than the register-resident kernel holds, not whole chunks for the ring kernel -> the tile kernel
    assert not _persistent(_desc(32, 224, 224, cs=[16, 16, 16], N=16))
    # a 14 x 14 map: too few tiles for the ring kernel to fill the machine
    assert not _persistent(_desc(32, 14, 14, cs=[256, 256], N=512))


@pytest.mark.parametrize("rows,K,N,mode", [(3152, 1024, 1024, 2), (3152, 1024, 3072, 1), (200704, 64, 256, 2), (50176, 128, 128, 2),
                                           (12544, 320, 1280, 2)])
def test_bias_gradient_modes_of_the_token_weight_gradient(rows, K, N, mode):
    """1: the one-split launch writes the bias gradient; 2: the split mode writes one partial row per split (ChangeFormer's long linears)"""
    lib = _lib.load()
    x, dy, g = torch.empty((16,), dtype=BF), torch.empty((16,), dtype=BF), torch.empty((16,), dtype=torch.float32)
    dw, ws = make_wgrad([SrcSpec(x, K, k_real=K)], dy, N, 0, N, g, 1, K, 0, 0, 1, rows, 1, rows, 1, 1, 1, 1, 0, BF)
    dw.bias_grad = g.data_ptr()
    assert lib.ksmi_conv_wgrad_fuses_bias(C.byref(dw), DT[BF]) == mode
    assert (dw.nsplit == 1) == (mode == 1)


def _phase_desc(B, H, W, Cin, N):
    """one 2 x 2 stride-1 phase convolution of ConvTranspose2d(k4, s2, p1) as plan_base._deconv builds it: strided output view"""
    x = torch.empty((16,), dtype=BF)
    src = SrcSpec(x, Cin)
    src.C = Cin
    out = torch.empty((16,), dtype=BF)
    d, _ = make_conv([src], [(out, N, 0, 0, N, 0)], out, None, None, B, H, W, H, W, 2, 2, 1, 1, N, BF, pad_x=1,
                     out_map=(2, 2, 0, 0, 2 * H, 2 * W))
    d._keep = (x, out)
    return d


@pytest.mark.parametrize("H,N", [(56, 256), (64, 256), (28, 128), (14, 128), (112, 128)])
def test_every_accepted_phase_geometry_has_an_instance(H, N):
    """for the 2 x 2 phase convolutions ksmi_igemm4_geom listed 64-column tile candidates that have no compiled
    2 x 2 instance; in the window 64 <= tiles < 192 / gy (ragged last batches, small data-parallel shards) the forward raised
    KSMI_E_UNSUPPORTED instead of falling back.  Sweep the batch: whatever kernel the dispatcher picks must be launchable, and a
    persistent-kernel choice for a phase convolution is the <4, 4> x 8-wave instance."""
    lib = _lib.load()
    info = (C.c_int32 * 8)()
    seen = set()
    for B in range(1, 49):
        d = _phase_desc(B, H, H, 256, N)
        assert lib.ksmi_conv_dispatch_info(C.byref(d), DT[BF], info) == 0
        gen, ok = info[0], info[1]
        assert ok == 1, (B, H, N, list(info))
        if gen == 4:
            assert (info[2], info[3], info[4]) == (4, 4, 8), (B, list(info))
        seen.add(gen)
    assert seen <= {2, 4}


def test_every_accepted_3x3_geometry_has_an_instance():
    """the same sweep over the 3 x 3 layers of the plans: batch 1 .. 40 at every SNUNet level and column count, plain / mask / long-K"""
    lib = _lib.load()
    info = (C.c_int32 * 8)()
    for H, cs, N in ((224, [32, 32, 64], 32), (112, [64, 64, 128], 64), (56, [128, 128, 256], 128), (28, [256, 256, 512], 256),
                     (14, [512], 512), (224, [256], 256), (224, [256], 2), (56, [64], 128), (112, [64], 64)):
        for B in (1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 40):
            for mask in (False, True):
                if mask and N < 16:
                    continue
                d = _desc(B, H, H, cs, N, out_c=(8 if N < 8 else None), mask=mask)
                assert lib.ksmi_conv_dispatch_info(C.byref(d), DT[BF], info) == 0
                assert info[1] == 1, (B, H, cs, N, mask, list(info))
