import os, sys, torch
sys.path.insert(0, "/root/repo")
from kurosiwo_amd import functional as Fk
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, H, cs, N in [("K=32", 224, [32], 32), ("K=192", 224, [32]*5+[64], 32), ("L1 K=256", 112, [64,64,128], 64)]:
    xs = [(torch.randn(32, H, H, c, device=dev) * 0.5).to(torch.bfloat16) for c in cs]
    w = torch.randn(N, sum(cs), 3, 3, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    for stats in (False, True):
        for bias in (None, b):
            t = timeit(lambda: Fk.conv3x3(xs, w, bias, want_stats=stats))
            print(f"{name} dbg={os.environ.get('KSMI_DBG','0')} stats={stats} bias={bias is not None}: {t*1e3:.1f} us")
