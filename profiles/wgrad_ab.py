"""Phase switches of the weight-gradient kernel (KSMI_WDBG: 1 no MFMA, 2 no global loads, 4 no LDS stores)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kurosiwo_amd import functional as Fk
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, H, cs, N in [("L0 K=32 N=32", 224, [32], 32), ("L0 K=224 N=32", 224, [32] * 5 + [64], 32), ("L1 K=256 N=64", 112, [64, 64, 128], 64), ("L2 K=128 N=128", 56, [128], 128)]:
    xs = [(torch.randn(32, H, H, c, device=dev) * 0.5).to(torch.bfloat16) for c in cs]
    dy = (torch.randn(32, H, H, N, device=dev) * 0.5).to(torch.bfloat16)
    t = timeit(lambda: Fk.conv3x3_wgrad(xs, dy))
    print(f"{name} wdbg={os.environ.get('KSMI_WDBG', '0')}: {t * 1e3:.1f} us")
