#!/usr/bin/env python3
"""Per-shape A/B of the long-K convolution kernels at the SNUNet bs=32 layer shapes: run once with KSMI_IGEMM4_OFF=1 (igemm2.hip)
and once without (igemm4.hip); prints us / TFLOP/s / algorithmic GB/s per shape.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kurosiwo_amd import functional as Fk  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("MB_B", "32"))
DT = torch.bfloat16


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(DT)


SHAPES = [  # name, H, cs, N, mode
    ("L0 conv2 32->32 aff", 224, [32], 32, "aff"),
    ("L0 dgrad2 32->32 mask", 224, [32], 32, "mask"),
    ("L0 conv0_1 128->32", 224, [32, 32, 64], 32, ""),
    ("L0 conv0_4 224->32", 224, [32] * 5 + [64], 32, ""),
    ("L0 dx0_0 128->32", 224, [32] * 4, 32, ""),
    ("X0 one source 224->32", 224, [224], 32, ""),            # over-fetch experiments: the same K as conv0_4 from one / seven tensors
    ("X0 six x32 192->32", 224, [32] * 6, 32, ""),
    ("X0 three x64 192->32", 224, [64] * 3, 32, ""),
    ("X0 one x192 192->32", 224, [192], 32, ""),
    ("X1 two x32 64->64", 112, [32, 32], 64, ""),
    ("X1 one x64 64->64", 112, [64], 64, ""),
    ("L1 conv2 64->64 aff", 112, [64], 64, "aff"),
    ("L1 dgrad2 64->64 mask", 112, [64], 64, "mask"),
    ("L1 conv1_1 256->64", 112, [64, 64, 128], 64, ""),
    ("L1 conv1_3 384->64", 112, [64] * 4 + [128], 64, ""),
    ("L1 dUp2 64->128", 112, [64], 128, ""),
    ("L2 conv2 128->128 aff", 56, [128], 128, "aff"),
    ("L2 dgrad2 128->128 mask", 56, [128], 128, "mask"),
    ("L2 conv2_1 512->128", 56, [128, 128, 256], 128, ""),
    ("L2 conv2_2 640->128", 56, [128] * 3 + [256], 128, ""),
    ("L2 dUp3 128->256", 56, [128], 256, ""),
    ("L3 conv2 256->256 aff", 28, [256], 256, "aff"),
    ("L3 dgrad2 256->256 mask", 28, [256], 256, "mask"),
    ("L3 conv3_1 1024->256", 28, [256, 256, 512], 256, ""),
    ("L4 conv2 512->512 aff", 14, [512], 512, "aff"),
    ("L4 conv4_0.c1 256->512", 14, [256], 512, ""),
]
which = os.environ.get("MB_ONLY", "")
for name, H, cs, N, mode in SHAPES:
    if which and which not in name:
        continue
    xs = [rnd(B, H, H, c) for c in cs]
    K = sum(cs)
    w = torch.randn(N, K, 3, 3, device=dev) * 0.05
    px = B * H * H
    fl = 2 * px * N * K * 9
    aff = (torch.ones(K, device=dev), torch.zeros(K, device=dev), 1) if mode == "aff" else None
    mask = None
    if mode == "mask":
        mask = (rnd(B, H, H, N), torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev))
    out = torch.empty((B, H, H, N), dtype=DT, device=dev)
    # build the descriptor once (weights packed once), time the launch alone
    import ctypes as C
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import DT as DTM, SrcSpec, conv_stats_rows, make_conv, stream_ptr
    srcs = [SrcSpec(x, x.shape[3]) for x in xs]
    if aff is not None:
        srcs[0].scale, srcs[0].shift, srcs[0].relu = aff
    d, table = make_conv(srcs, [(out, N, 0, 0, N, 0)], out, None, None, B, H, H, H, H, 3, 3, 1, 1, N, DT, mask=mask)
    wpk = Fk._pack(w.contiguous(), table, 9, N, N, 9, K * 9, 0, 1, 0, DT)
    d.wpk = wpk.data_ptr()
    stats = torch.zeros((conv_stats_rows(d, DT), 2, d.Npad), dtype=torch.float32, device=dev)
    d.stats = stats.data_ptr()
    lib = _lib.load()
    ms = timeit(lambda: _lib.check(lib.ksmi_conv_forward(C.byref(d), DTM[DT], stream_ptr()), "conv"))
    nbytes = (px * K + px * N * (2 if mask else 1)) * 2
    if int(os.environ.get("KSMI_IG4_DBG", "0")) & 128:
        st = stats.view(torch.int64).flatten()[:8].cpu().tolist()
        print("   stamps (cycles from start):", [x - st[0] for x in st if x])
    print(f"{name:28s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:8.1f} TF/s  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)
