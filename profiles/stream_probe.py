"""Measured device memory roofline next to the 8 TB/s spec figure: read, copy and triad streams over buffers
far larger than L2 + MALL, and a cache-resident copy for contrast (torch elementwise kernels; sizes in MiB of one operand)."""
import torch
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mib in (32, 128, 1024, 4096):
    n = mib * (1 << 20) // 4
    a = torch.empty(n, device=dev); b = torch.randn(n, device=dev); c = torch.randn(n, device=dev)
    t_copy = timeit(lambda: a.copy_(b))
    t_triad = timeit(lambda: torch.add(b, c, alpha=2.0, out=a))
    t_read = timeit(lambda: b.sum())
    t_fill = timeit(lambda: a.fill_(1.0))
    bts = n * 4
    print(f"{mib} MiB/operand: read {bts / t_read / 1e12:.2f} TB/s  fill {bts / t_fill / 1e12:.2f}  copy {2 * bts / t_copy / 1e12:.2f}  triad {3 * bts / t_triad / 1e12:.2f}")
