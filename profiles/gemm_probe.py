"""Token-GEMM shapes of FloodViT (rows 3152) and ChangeFormer: libksmi GEMM kernels next to torch.matmul (hipBLASLt / rocBLAS)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kurosiwo_amd import functional as Fk
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rows, K, N in [(3152, 1024, 3072), (3152, 1024, 1024), (3152, 1024, 2048), (3152, 2048, 1024), (100352, 64, 64), (25088, 128, 512), (6272, 320, 1280)]:
    x = (torch.randn(rows, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    dy = (torch.randn(rows, N, device=dev) * 0.5).to(torch.bfloat16)
    fl = 2.0 * rows * K * N
    t_nt = timeit(lambda: Fk.gemm_nt(x, w))
    t_nn = timeit(lambda: Fk.gemm_nn(dy, w))
    t_wg = timeit(lambda: Fk.linear_wgrad(x, dy))
    t_tnt = timeit(lambda: torch.matmul(x, w.t()))
    t_tnn = timeit(lambda: torch.matmul(dy, w))
    t_twg = timeit(lambda: torch.matmul(dy.t(), x))
    t_twg32 = timeit(lambda: torch.matmul(dy.t(), x).float())
    print(f"rows {rows} K {K} N {N}: ksmi nt {t_nt*1e3:.1f} us ({fl/t_nt/1e9:.0f} TF/s) nn {t_nn*1e3:.1f} ({fl/t_nn/1e9:.0f}) wgrad {t_wg*1e3:.1f} ({fl/t_wg/1e9:.0f}) | "
          f"torch nt {t_tnt*1e3:.1f} ({fl/t_tnt/1e9:.0f}) nn {t_tnn*1e3:.1f} ({fl/t_tnn/1e9:.0f}) tn {t_twg*1e3:.1f} ({fl/t_twg/1e9:.0f}) tn+f32 {t_twg32*1e3:.1f}")
