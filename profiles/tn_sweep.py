"""nn.Linear weight gradient (gemm2_tn) over its tiling choices at the FloodViT shapes: B-side tile 128 / 96 / 64 x row splits.
One subprocess per choice (KSMI_TN_BT / KSMI_TN_SPLIT are read once); descriptor built once, the launch (+ reducer) timed alone.
  python profiles/tn_sweep.py            -> the table
  python profiles/tn_sweep.py one        -> one line for the current environment (the chooser's own pick without the variables)"""
import ctypes as C
import os
import subprocess
import sys

SHAPES = [(3152, 1024, 3072), (3152, 1024, 1024), (3152, 1024, 2048), (3152, 2048, 1024), (6272, 320, 1280), (25088, 128, 512)]


def one():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import DT, SrcSpec, make_wgrad, stream_ptr
    dev = torch.device("cuda:0")
    lib = _lib.load()
    out = []
    for rows, K, N in SHAPES:
        torch.manual_seed(1)
        x = (torch.randn(rows, K, device=dev) * 0.5).bfloat16()
        dy = (torch.randn(rows, N, device=dev) * 0.5).bfloat16()
        grad = torch.zeros((N, K), dtype=torch.float32, device=dev)
        d, ws = make_wgrad([SrcSpec(x, K)], dy, N, 0, N, grad, 1, K, 0, 0, 1, rows, 1, rows, 1, 1, 1, 1, 0, torch.bfloat16)
        wsb = torch.empty(max(ws, 16), dtype=torch.uint8, device=dev)
        d.partial = wsb.data_ptr()
        st = stream_ptr()

        def run():
            _lib.check(lib.ksmi_conv_wgrad(C.byref(d), DT[torch.bfloat16], st), "wgrad")
        for _ in range(3):
            run()
        ref = dy.float().t() @ x.float()
        err = ((grad - ref).norm() / ref.norm()).item()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        out.append(f"{us:.1f}/{d.nsplit}" + ("" if err < 1e-5 else f"!err {err:.1e}"))
    print(" ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one()
        sys.exit(0)
    print("us/nsplit per shape (rows,K,N): " + " ".join(str(s) for s in SHAPES), flush=True)
    for bt in (0, 128, 96, 64):
        for s in ((0,) if bt == 0 else (1, 2, 3, 4, 6, 8)):
            env = dict(os.environ)
            if bt:
                env["KSMI_TN_BT"], env["KSMI_TN_SPLIT"] = str(bt), str(s)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
            print(f"bt {bt or 'auto':>4} split {s or 'auto':>4}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
