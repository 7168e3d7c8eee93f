"""gemm2 (nn.Linear forward / input gradient) over row-tile height MT x LDS ring depth NS at the FloodViT shapes, next to torch.matmul;
and the weight-gradient kernel's ring depth.  One subprocess per choice (the variables are read once).
  python profiles/gemm2_sweep.py        -> the table        python profiles/gemm2_sweep.py one -> one line for this environment"""
import os
import subprocess
import sys

SHAPES = [(3152, 1024, 3072), (3152, 1024, 1024), (3152, 1024, 2048), (3152, 2048, 1024)]


def timeit(fn, n=50):
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def one(with_torch):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import stream_ptr
    dev = torch.device("cuda:0")
    lib = _lib.load()
    out = []
    for rows, K, N in SHAPES:
        torch.manual_seed(1)
        x = (torch.randn(rows, K, device=dev) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        dy = (torch.randn(rows, N, device=dev) * 0.5).bfloat16()
        y = torch.empty(rows, N, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(rows, K, dtype=torch.bfloat16, device=dev)
        st = stream_ptr()
        nt = lambda: lib.ksmi_gemm_nt(x.data_ptr(), K, w.data_ptr(), K, None, None, N, y.data_ptr(), N, rows, K, N, st)
        nn = lambda: lib.ksmi_gemm_nn(dy.data_ptr(), N, w.data_ptr(), K, dx.data_ptr(), K, rows, K, N, 0, st)
        if with_torch:
            out.append(f"{timeit(lambda: torch.matmul(x, w.t(), out=y)):.1f}|{timeit(lambda: torch.matmul(dy, w, out=dx)):.1f}")
            continue
        assert nt() == 0 and nn() == 0
        e1 = ((y.float() - x.float() @ w.float().t()).norm() / y.float().norm()).item()
        e2 = ((dx.float() - dy.float() @ w.float()).norm() / dx.float().norm()).item()
        out.append(f"{timeit(nt):.1f}|{timeit(nn):.1f}" + ("" if max(e1, e2) < 4e-3 else f"!err {e1:.1e} {e2:.1e}"))
    print(" ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[1] == "torch")
        sys.exit(0)
    me = os.path.abspath(__file__)
    print("forward|input-gradient us per shape (rows,K,N): " + " ".join(str(s) for s in SHAPES), flush=True)
    r = subprocess.run([sys.executable, me, "torch"], capture_output=True, text=True)
    print(f"torch.matmul   : {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
    nss = tuple(int(v) for v in os.environ.get("SWEEP_NS", "2,3,4,5").split(","))
    for ns in nss:
        for mt in (0, 2, 3, 4, 5, 6, 7, 8):
            env = dict(os.environ, KSMI_GEMM2_NS=str(ns))
            if mt:
                env["KSMI_GEMM2_MT"] = str(mt)
            r = subprocess.run([sys.executable, me, "one"], env=env, capture_output=True, text=True)
            print(f"NS {ns} MT {mt or 'auto':>4}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
    tn = os.path.join(os.path.dirname(me), "tn_sweep.py")
    for ns in (3, 4, 5, 6) if "SWEEP_NS" not in os.environ else ():
        for bt in (0, 128, 96, 64):
            env = dict(os.environ, KSMI_TN_NS=str(ns))
            if bt:
                env["KSMI_TN_BT"], env["KSMI_TN_SPLIT"] = str(bt), "1"
            r = subprocess.run([sys.executable, tn, "one"], env=env, capture_output=True, text=True)
            print(f"wgrad NS {ns} bt {bt or 'auto':>4} direct: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
