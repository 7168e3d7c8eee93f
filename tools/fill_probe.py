"""LDS-fill rate of the part by where the data lives (GPU box): ksmi_hbm_probe kind 0 (LDS-DMA reads, 512 workgroups x 8 waves x 8 KB in
flight) with the addresses wrapping inside a window.  usage: python tools/fill_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from kurosiwo_amd import _lib
from kurosiwo_amd.runtime import stream_ptr
lib = _lib.load()
dev = torch.device("cuda:0")
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
def run(mode, reps=3):
    best = None
    for it in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.ksmi_hbm_probe(mode, a.data_ptr(), None, None, n, sink.data_ptr(), stream_ptr()), "probe")
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1)
        if it and (best is None or ms < best):
            best = ms
    return n / best / 1e6
print(f"stream 1 GiB from HBM                      : {run(0):8.1f} GB/s")
for wlog in (16, 18, 20, 22, 24, 26, 28):
    for shared in (0, 1):
        print(f"window {(1 << wlog) >> 10:8d} KiB, {'same addresses in every workgroup' if shared else 'disjoint slices per workgroup  '}: {run(wlog << 8 | shared << 14):8.1f} GB/s")
