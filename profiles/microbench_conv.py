#!/usr/bin/env python3
"""Per-shape micro-benchmark of the implicit-GEMM family at the SNUNet bs=32 layer shapes
(optimisation inner loop; run on the GPU box).  Prints ms / TFLOP/s / algorithmic GB/s."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kurosiwo_amd import functional as Fk  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("MB_B", "32"))
DT = torch.bfloat16 if os.environ.get("MB_DT", "bf16") == "bf16" else torch.float32
ES = 2 if DT == torch.bfloat16 else 4


def timeit(fn, n=5):
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


def report(name, ms, flops, nbytes):
    print(f"{name:34s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TF/s  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)


SHAPES = [  # name, H, cs, N
    ("L0 conv0_1.conv1 128->32", 224, [32, 32, 64], 32),
    ("L0 conv0_4.conv1 224->32", 224, [32] * 5 + [64], 32),
    ("L0 conv2 32->32 (affine)", 224, [32], 32),
    ("L1 conv1_1.conv1 256->64", 112, [64, 64, 128], 64),
    ("L1 conv2 64->64 (affine)", 112, [64], 64),
    ("L2 conv2_1.conv1 512->128", 56, [128, 128, 256], 128),
    ("L3 conv3_1.conv1 1024->256", 28, [256, 256, 512], 256),
    ("L4 conv4_0.conv2 512->512", 14, [512], 512),
]
which = os.environ.get("MB_ONLY", "")
for name, H, cs, N in SHAPES:
    if which and which not in name:
        continue
    xs = [rnd(B, H, H, c) for c in cs]
    K = sum(cs)
    w = torch.randn(N, K, 3, 3, device=dev) * 0.05
    dy = rnd(B, H, H, N)
    px = B * H * H
    fl = 2 * px * N * K * 9
    aff = (torch.ones(cs[0], device=dev), torch.zeros(cs[0], device=dev), 1) if "affine" in name else None
    ms = timeit(lambda: Fk.conv3x3(xs, w, None, affine=aff, want_stats=True))
    report("fwd   " + name, ms, fl, (px * K + px * N) * ES)
    ms = timeit(lambda: Fk.conv3x3_dgrad(dy, w, cs))
    report("dgrad " + name, ms, fl, (px * K + px * N) * ES)
    ms = timeit(lambda: Fk.conv3x3_wgrad(xs, dy, affine=aff))
    report("wgrad " + name, ms, fl, (px * K + px * N) * ES)
for name, H, Cc in [("up 64 @112->224", 112, 64), ("up 128 @56->112", 56, 128), ("up 512 @14->28", 14, 512)]:
    if which and which not in name:
        continue
    x = rnd(B, H, H, Cc)
    w = torch.randn(Cc, Cc, 2, 2, device=dev) * 0.05
    bias = torch.zeros(Cc, device=dev)
    dy = rnd(B, 2 * H, 2 * H, Cc)
    px = B * H * H
    fl = 2 * px * Cc * Cc * 4
    ms = timeit(lambda: Fk.deconv2x2(x, w, bias))
    report("fwd   " + name, ms, fl, (px * Cc + 4 * px * Cc) * ES)
    ms = timeit(lambda: Fk.deconv2x2_backward(x, dy, w))
    report("bwd   " + name, ms, 2 * fl, 2 * (px * Cc + 4 * px * Cc) * ES)
