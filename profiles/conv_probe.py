"""Per-shape timing of the short-K convolution launches of the SNUNet bs=32 step (forward conv with statistics, fused BN-apply
operand, ConvTranspose forward as 1x1 + pixel shuffle, its input gradient as 2x2 stride 2): persistent kernel (csrc/igemm3.hip)
vs igemm2 (KSMI_IGEMM3_OFF=1 in a second process).  Prints microseconds, TFLOP/s and algorithmic GB/s.

    python profiles/conv_probe.py ; KSMI_IGEMM3_OFF=1 python profiles/conv_probe.py
"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kurosiwo_amd import _lib
from kurosiwo_amd.runtime import DT, SrcSpec, conv_stats_rows, make_conv, make_pack, packed_weight_numel, stream_ptr

dt = torch.bfloat16


def pack(w, table, taps, N, n_mod, sK, sN, sD, sT, flip):
    Npad = (N + 15) // 16 * 16
    out = torch.empty(packed_weight_numel(table, taps, Npad, dt), dtype=dt, device=w.device)
    d = make_pack(w, out, table, taps, N, Npad, n_mod, sK, sN, sD, sT, flip)
    _lib.check(_lib.load().ksmi_pack_weights(C.byref(d), DT[dt], stream_ptr()))
    return out


def timeit(d, n=20):
    lib = _lib.load()
    for _ in range(3):
        _lib.check(lib.ksmi_conv_forward(C.byref(d), DT[dt], stream_ptr()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lib.ksmi_conv_forward(C.byref(d), DT[dt], stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = 32
    dev = torch.device("cuda:0")
    keep = []
    rows = []
    for (H, K, N, aff) in [(224, 32, 32, True), (224, 32, 32, False), (224, 32, 64, False), (112, 64, 64, True), (112, 64, 64, False),
                           (112, 32, 64, False), (112, 64, 128, False), (56, 64, 128, False)]:
        x = torch.randn((B, H, H, K), device=dev).to(dt)
        y = torch.empty((B, H, H, N), dtype=dt, device=dev)
        w = torch.randn((N, K, 3, 3), device=dev) * 0.05
        bias = torch.zeros(N, device=dev)
        s = SrcSpec(x, K)
        if aff:
            s.scale, s.shift, s.relu = torch.ones(K, device=dev), torch.zeros(K, device=dev), 1
        d, table = make_conv([s], [(y, N, 0, 0, N, 0)], y, bias, None, B, H, H, H, H, 3, 3, 1, 1, N, dt)
        wp = pack(w, table, 9, N, N, 9, K * 9, 0, 1, 0)
        d.wpk = wp.data_ptr()
        r = conv_stats_rows(d, dt)
        st = torch.zeros((r, 2, d.Npad), device=dev)
        d.stats = st.data_ptr()
        keep += [x, y, w, bias, wp, st, s.scale, s.shift]
        ms = timeit(d)
        fl, by = 2.0 * B * H * H * N * K * 9, B * H * H * (K + N) * 2
        print(f"3x3 H={H:4d} K={K:4d} N={N:4d} aff={int(aff)} rows={r:5d} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s {by / ms / 1e6:7.1f} GB/s", flush=True)
    for (H, Cc) in [(112, 64), (56, 128), (28, 256)]:
        x = torch.randn((B, H, H, Cc), device=dev).to(dt)
        y = torch.empty((B, 2 * H, 2 * H, Cc), dtype=dt, device=dev)
        w = torch.randn((Cc, Cc, 2, 2), device=dev) * 0.05
        bias = torch.zeros(Cc, device=dev)
        d, table = make_conv([SrcSpec(x, Cc)], [(y, Cc, 0, 0, 4 * Cc, 0)], x, bias, None, B, H, H, H, H, 1, 1, 1, 0, 4 * Cc, dt, ps_cout=Cc)
        wp = pack(w, table, 1, 4 * Cc, Cc, Cc * 4, 4, 1, 0, 0)
        d.wpk = wp.data_ptr()
        ms = timeit(d)
        by = B * H * H * Cc * 2 * 5
        print(f"up-fwd (1x1 -> 4C) H={H:4d} C={Cc:4d} {ms * 1e3:8.1f} us {by / ms / 1e6:7.1f} GB/s", flush=True)
        dx = torch.empty_like(x)
        d2, t2 = make_conv([SrcSpec(y, Cc)], [(dx, Cc, 0, 0, Cc, 0)], y, None, None, B, 2 * H, 2 * H, H, H, 2, 2, 2, 0, Cc, dt)
        w2 = pack(w, t2, 4, Cc, Cc, 4, Cc * 4, 0, 1, 0)
        d2.wpk = w2.data_ptr()
        ms = timeit(d2)
        print(f"up-dgrad (2x2 s2)    H={H:4d} C={Cc:4d} {ms * 1e3:8.1f} us {by / ms / 1e6:7.1f} GB/s", flush=True)
        keep += [x, y, w, bias, wp, dx, w2]


if __name__ == "__main__":
    main()
