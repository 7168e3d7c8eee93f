"""Would pre-transposed operands pay for the ViT-size weight gradient?  dW[N][K] = dY^T X as (a) the direct product on row-major
operands (reduction dimension strided in both: what ksmi_lt_linear_wgrad asks hipBLASLt for) and (b) transposes + the product with
the reduction dimension contiguous in both operands (torch ops: hipBLASLt / elementwise copy kernels)."""
import torch
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, K, N in [(3152, 1024, 3072), (3152, 1024, 1024), (3152, 1024, 2048), (3152, 2048, 1024), (6272, 512, 2048)]:
    x = (torch.randn(rows, K, device=dev) * 0.5).bfloat16()
    dy = (torch.randn(rows, N, device=dev) * 0.5).bfloat16()
    xt, dyt = x.t().contiguous(), dy.t().contiguous()
    out32 = torch.empty(N, K, device=dev)
    t_direct = timeit(lambda: torch.matmul(dy.t(), x))
    t_tn = timeit(lambda: torch.matmul(dyt, xt.t()))
    t_tr = timeit(lambda: (x.t().contiguous(), dy.t().contiguous()))
    print(f"rows {rows} K {K} N {N}: direct {t_direct:.1f} us | contiguous-reduction product {t_tn:.1f} + transposes {t_tr:.1f} = {t_tn + t_tr:.1f} us")
