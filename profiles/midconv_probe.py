"""Phase breakdown of the long-K 3x3 convolutions (igemm2): time with KSMI_DBG = 0 / 1 (no MFMA) / 2 (no DMA of the next chunks) / 4 (no
epilogue) on SNUNet's mid layers and ChangeFormer's 256-channel layers.  KSMI_DBG is read once per process:
    for d in 0 1 2 4; do KSMI_DBG=$d python profiles/midconv_probe.py; done
"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kurosiwo_amd import _lib
from kurosiwo_amd.runtime import DT, SrcSpec, make_conv
from profiles.conv_probe import pack, timeit

dt = torch.bfloat16


def main():
    dev = torch.device("cuda:0")
    B = 32
    keep = []
    for (H, K, N) in [(112, 256, 64), (112, 384, 64), (56, 512, 128), (28, 1024, 256), (224, 256, 256), (112, 256, 256), (224, 224, 32)]:
        x = torch.randn((B, H, H, K), device=dev).to(dt)
        y = torch.empty((B, H, H, N), dtype=dt, device=dev)
        w = torch.randn((N, K, 3, 3), device=dev) * 0.02
        bias = torch.zeros(N, device=dev)
        d, table = make_conv([SrcSpec(x, K)], [(y, N, 0, 0, N, 0)], y, bias, None, B, H, H, H, H, 3, 3, 1, 1, N, dt)
        wp = pack(w, table, 9, N, N, 9, K * 9, 0, 1, 0)
        d.wpk = wp.data_ptr()
        keep += [x, y, w, bias, wp]
        ms = timeit(d, n=10)
        fl = 2.0 * B * H * H * N * K * 9
        print(f"DBG={os.environ.get('KSMI_DBG', '0')} 3x3 H={H:4d} K={K:4d} N={N:4d} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
