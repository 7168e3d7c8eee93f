"""Per-shape timing of the 3x3 weight gradient at the SNUNet bs=32 shapes: channel-owner kernel (csrc/wgrad3.hip) vs the first
kernel (KSMI_WGRAD3_OFF=1 in a second process).  Prints TFLOP/s and algorithmic GB/s per launch (kernel + slab reducer).

    python profiles/wgrad3_probe.py            # new kernel
    KSMI_WGRAD3_OFF=1 python profiles/wgrad3_probe.py
"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kurosiwo_amd import _lib
from kurosiwo_amd.runtime import DT, SrcSpec, make_wgrad, stream_ptr

SHAPES = [  # (H, K list, N)
    (224, [32], 32), (224, [32, 32, 64], 32), (224, [32, 32, 32, 32, 32, 64], 32),
    (112, [32], 64), (112, [64], 64), (112, [64, 64, 128], 64), (112, [64, 64, 64, 64, 128], 64),
    (56, [64], 128), (56, [128], 128), (56, [128, 128, 256], 128), (56, [128, 128, 128, 256], 128),
    (28, [128], 256), (28, [256], 256), (28, [256, 256, 512], 256),
    (14, [256], 512), (14, [512], 512),
]


def main():
    B = int(os.environ.get("PROBE_B", "32"))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    dt = torch.bfloat16
    tot = 0.0
    for H, cs, N in SHAPES:
        xs = [torch.randn((B, H, H, c), device=dev).to(dt) for c in cs]
        dy = torch.randn((B, H, H, N), device=dev).to(dt)
        K = sum(cs)
        grad = torch.zeros((N, K, 3, 3), dtype=torch.float32, device=dev)
        d, ws = make_wgrad([SrcSpec(x, x.shape[3]) for x in xs], dy, N, 0, N, grad, 9, K * 9, 1, 0, B, H, H, H, H, 3, 3, 1, 1, dt)
        wsb = torch.empty(max(ws, 16), dtype=torch.uint8, device=dev)
        d.partial = wsb.data_ptr()
        for _ in range(3):
            _lib.check(lib.ksmi_conv_wgrad(C.byref(d), DT[dt], stream_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            lib.ksmi_conv_wgrad(C.byref(d), DT[dt], stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * H * H * N * K * 9
        by = B * H * H * (K + N) * 2
        tot += ms
        print(f"H={H:4d} K={K:5d} N={N:4d} nsplit={d.nsplit:4d} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s {by / ms / 1e6:7.1f} GB/s", flush=True)
    print(f"sum {tot:.3f} ms")


if __name__ == "__main__":
    main()
