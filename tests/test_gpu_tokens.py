"""GPU parity of the token-sequence kernels of the FloodViT path (through the C-ABI) against stock
torch-CPU fp32 ops on the same seeded inputs (rows V1-V4 of SURVEY.md §8(a))."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle.seeded import seeded_tensor


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _tol(dtype):
    return 3e-5 if dtype == torch.float32 else 3e-2


def q(t, dtype):
    return t.to(dtype).float()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cin,n", [(394, 1024, 3072), (197, 1536, 1024), (300, 2048, 1024), (64, 64, 48)])
def test_linear_forward_dgrad_wgrad(dev, dtype, rows, cin, n):
    from kurosiwo_amd import functional as Fk
    tag = f"lin{rows}{cin}{n}"
    x = seeded_tensor(tag + "x", (rows, cin))
    w = seeded_tensor(tag + "w", (n, cin)) * cin ** -0.5
    b = seeded_tensor(tag + "b", (n,)) * 0.1
    dy = seeded_tensor(tag + "dy", (rows, n))
    xr, wr = q(x, dtype).requires_grad_(True), q(w, dtype).requires_grad_(True)
    y_ref = F.linear(xr, wr, b)
    y_ref.backward(q(dy, dtype))
    xd, dyd = x.to(dev).to(dtype), dy.to(dev).to(dtype)
    y = Fk.linear(xd, w.to(dev), b.to(dev))
    tol = _tol(dtype)
    assert (y.float().cpu() - y_ref.detach()).abs().max() < tol * y_ref.abs().max()
    dx = Fk.linear_dgrad(dyd, w.to(dev))
    assert (dx.float().cpu() - xr.grad).abs().max() < tol * xr.grad.abs().max()
    dw = Fk.linear_wgrad(xd, dyd)
    assert (dw.cpu() - wr.grad).abs().max() < tol * wr.grad.abs().max()
    # residual form: out += x @ W^T
    acc = Fk.linear(xd, w.to(dev), None, out=y.clone(), accumulate=1)
    ref2 = q(y.float().cpu(), dtype) + F.linear(q(x, dtype), q(w, dtype))
    assert (acc.float().cpu() - ref2).abs().max() < 2 * tol * ref2.abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,c", [(394, 1024), (197, 1536), (37, 64)])
def test_layernorm(dev, dtype, rows, c):
    from kurosiwo_amd import functional as Fk
    tag = f"ln{rows}{c}"
    x = seeded_tensor(tag + "x", (rows, c)) * 2 + 0.5
    g = 1 + 0.2 * seeded_tensor(tag + "g", (c,))
    b = 0.1 * seeded_tensor(tag + "b", (c,))
    dy = seeded_tensor(tag + "dy", (rows, c))
    xr, gr, br = q(x, dtype).requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.layer_norm(xr, (c,), gr, br, 1e-5)
    y_ref.backward(q(dy, dtype))
    xd = x.to(dev).to(dtype)
    y, mean, rstd = Fk.layernorm(xd, g.to(dev), b.to(dev))
    tol = _tol(dtype)
    assert (y.float().cpu() - y_ref.detach()).abs().max() < tol * y_ref.abs().max()
    dx, dg, db = Fk.layernorm_backward(dy.to(dev).to(dtype), xd, mean, rstd, g.to(dev))
    assert (dx.float().cpu() - xr.grad).abs().max() < tol * xr.grad.abs().max()
    assert (dg.cpu() - gr.grad).abs().max() < 1e-3 * gr.grad.abs().max()
    assert (db.cpu() - br.grad).abs().max() < 1e-3 * br.grad.abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gelu(dev, dtype):
    from kurosiwo_amd import functional as Fk
    x = seeded_tensor("gelu.x", (100, 64)) * 3
    dy = seeded_tensor("gelu.dy", (100, 64))
    xr = q(x, dtype).requires_grad_(True)
    y_ref = F.gelu(xr)
    y_ref.backward(q(dy, dtype))
    xd = x.to(dev).to(dtype)
    tol = _tol(dtype)
    assert (Fk.gelu(xd).float().cpu() - y_ref.detach()).abs().max() < tol * y_ref.abs().max()
    assert (Fk.gelu_backward(dy.to(dev).to(dtype), xd).float().cpu() - xr.grad).abs().max() < tol * xr.grad.abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,H", [(2, 197, 16), (1, 50, 2), (3, 300, 4)])
def test_attention(dev, dtype, B, N, H):
    from kurosiwo_amd import functional as Fk
    D = 64
    tag = f"att{B}{N}{H}"
    qkv = seeded_tensor(tag + "qkv", (B * N, 3 * H * D))
    dout = seeded_tensor(tag + "do", (B * N, H * D))
    qr = q(qkv, dtype).requires_grad_(True)
    t = qr.view(B, N, 3, H, D)
    qq, kk, vv = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))           # b h n d  ("(3 h d)" chunk order)
    attn = torch.softmax(qq @ kk.transpose(-1, -2) * D ** -0.5, dim=-1)
    o_ref = (attn @ vv).permute(0, 2, 1, 3).reshape(B * N, H * D)
    o_ref.backward(q(dout, dtype))
    qd = qkv.to(dev).to(dtype)
    out, lse = Fk.attention(qd, B, N, H)
    tol = _tol(dtype)
    assert (out.float().cpu() - o_ref.detach()).abs().max() < tol * o_ref.abs().max()
    dqkv = Fk.attention_backward(qd, out, lse, dout.to(dev).to(dtype), B, N, H)
    assert (dqkv.float().cpu() - qr.grad).abs().max() < tol * qr.grad.abs().max()


@pytest.mark.parametrize("rows,K,N", [(3152, 1024, 3072), (3152, 2048, 1024), (394, 1536, 1024), (200704, 64, 256), (98, 512, 2048), (50, 320, 40)])
def test_gemm_nt_nn_bf16(dev, rows, K, N):
    """128x128-tile token GEMMs vs torch (bf16 operands, fp32 accumulation)."""
    from kurosiwo_amd import functional as Fk
    torch.manual_seed(rows + K)
    x = (torch.randn(rows, K, device=dev) * 0.5).bfloat16()
    w = torch.randn(N, K, device=dev) * K ** -0.5
    b = torch.randn(N, device=dev)
    res = torch.randn(rows, N, device=dev).bfloat16()
    wb = Fk.cast_bf16(w)
    assert torch.equal(wb, w.bfloat16())
    ref = x.float() @ wb.float().t() + b
    y = Fk.gemm_nt(x, wb, b)
    assert float((y.float() - ref).abs().max() / ref.abs().max()) < 1e-2
    y2 = Fk.gemm_nt(x, wb, None, res)
    ref2 = x.float() @ wb.float().t() + res.float()
    assert float((y2.float() - ref2).abs().max() / ref2.abs().max()) < 1e-2
    dy = (torch.randn(rows, N, device=dev) * 0.5).bfloat16()
    refd = dy.float() @ wb.float()
    dx = Fk.gemm_nn(dy, wb)
    assert float((dx.float() - refd).abs().max() / refd.abs().max()) < 1e-2
    base = torch.randn(rows, K, device=dev).bfloat16()
    dx2 = Fk.gemm_nn(dy, wb, out=base.clone())
    assert float((dx2.float() - (refd + base.float())).abs().max() / refd.abs().max()) < 2e-2


# (rows, K, N): forward has N/128 column tiles, the input gradient K/128; ragged last row tiles, reductions of 1 .. 48 K steps
GEMM2_SHAPES = [(3152, 1024, 3072), (3152, 1024, 2048), (3152, 2048, 1024), (197, 128, 128), (4200, 192, 256), (5000, 256, 128),
                (3000, 64 * 5, 128 * 3), (2500, 448, 1280), (1000, 128, 640), (40000, 128, 128), (777, 1024, 256),
                (20000, 128, 128), (45000, 192, 128), (60000, 128, 128)]


def test_gemm2_every_row_tile_and_ring_depth():
    """Every instance gemm2_kernel<MT, ., NS> the chooser (pick_tile, csrc/gemm2.hip) can reach, and the probe-only ring depths: MT and
    NS pinned through KSMI_GEMM2_MT / KSMI_GEMM2_NS (read once per process), forward with bias + residual and input gradient with
    accumulation on ragged rows and on reductions shorter than, equal to and longer than the ring."""
    import os
    import subprocess
    import sys
    code = (
        "import torch\n"
        "from kurosiwo_amd import functional as Fk\n"
        "dev = torch.device('cuda:0')\n"
        "errs = []\n"
        "for rows, K, N in [(3152, 1024, 384), (777, 64, 128), (1000, 128, 256), (50, 192, 128), (4100, 320, 640)]:\n"
        "    torch.manual_seed(rows)\n"
        "    x = (torch.randn(rows, K, device=dev) * 0.5).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()\n"
        "    b = torch.randn(N, device=dev); res = torch.randn(rows, N, device=dev).bfloat16()\n"
        "    dy = (torch.randn(rows, N, device=dev) * 0.5).bfloat16(); base = torch.randn(rows, K, device=dev).bfloat16()\n"
        "    ref = x.float() @ w.float().t() + b + res.float(); refd = dy.float() @ w.float() + base.float()\n"
        "    errs.append(float((Fk.gemm_nt(x, w, b, res).float() - ref).abs().max() / ref.abs().max()))\n"
        "    errs.append(float((Fk.gemm_nn(dy, w, out=base.clone()).float() - refd).abs().max() / refd.abs().max()))\n"
        "print('ERR', max(errs))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    combos = [(mt, ns) for mt in range(2, 9) for ns in (2, 3)] + [(2, 5), (4, 4), (6, 4)]
    for mt, ns in combos:
        env = dict(os.environ, PYTHONPATH=root, KSMI_GEMM2_MT=str(mt), KSMI_GEMM2_NS=str(ns))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        assert float(out.stdout.split("ERR")[1]) < 2e-2, (mt, ns, out.stdout)


@pytest.mark.parametrize("rows,K,N", GEMM2_SHAPES)
def test_gemm2_lds_dma_tiles(dev, rows, K, N):
    """gemm2.hip (LDS-DMA ring, K step 64; the chooser's own pick per shape): forward with bias / residual, input gradient with and without accumulate,
    weight gradient (direct and split-slab modes) against torch on the same bf16 operands."""
    from kurosiwo_amd import functional as Fk
    torch.manual_seed(rows + K + N)
    x = (torch.randn(rows, K, device=dev) * 0.5).bfloat16()
    wb = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=dev)
    res = torch.randn(rows, N, device=dev).bfloat16()
    rel = lambda got, ref: float((got.float() - ref).abs().max() / ref.abs().max())
    ref = x.float() @ wb.float().t()
    assert rel(Fk.gemm_nt(x, wb, b), ref + b) < 1e-2
    assert rel(Fk.gemm_nt(x, wb, b, res), ref + b + res.float()) < 1e-2
    dy = (torch.randn(rows, N, device=dev) * 0.5).bfloat16()
    refd = dy.float() @ wb.float()
    assert rel(Fk.gemm_nn(dy, wb), refd) < 1e-2
    base = torch.randn(rows, K, device=dev).bfloat16()
    assert float((Fk.gemm_nn(dy, wb, out=base.clone()).float() - (refd + base.float())).abs().max() / refd.abs().max()) < 2e-2
    refw = dy.float().t() @ x.float()
    got = Fk.linear_wgrad(x, dy)
    assert got.dtype == torch.float32 and rel(got, refw) < 2e-3


@pytest.mark.parametrize("rows,K,N", [(3152, 1024, 3072), (3152, 1024, 1024), (3152, 2048, 1024), (1568, 512, 2048)])
def test_linear_wgrad_vit_size(dev, rows, K, N):
    """nn.Linear weight gradient at ViT size (gemm2_tn_kernel: direct fp32 write or split slabs + reducer): against torch on the
    same bf16 operands."""
    from kurosiwo_amd import functional as Fk
    torch.manual_seed(rows + N)
    x = (torch.randn(rows, K, device=dev) * 0.5).bfloat16()
    dy = (torch.randn(rows, N, device=dev) * 0.5).bfloat16()
    ref = dy.float().t() @ x.float()
    got = Fk.linear_wgrad(x, dy)
    assert got.dtype == torch.float32 and got.shape == (N, K)
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-3


def test_hand_written_gemm_generations_agree():
    """The ViT-size GEMMs on (a) the default product path (gemm2.hip: LDS-DMA tiles) and (b) the first-generation kernels
    (KSMI_GEMM2_OFF=1: gemm.hip / gemm_tn_wgrad_kernel).  The switch is read once per process: each variant runs in its own process and
    must agree with torch.  (The vendor-library comparison lives outside the product library: profiles/gemm_probe.py times torch.matmul
    = hipBLASLt next to these kernels.)"""
    import os
    import subprocess
    import sys
    code = (
        "import torch\n"
        "from kurosiwo_amd import functional as Fk\n"
        "torch.manual_seed(5)\n"
        "dev = torch.device('cuda:0')\n"
        "x = (torch.randn(3152, 1024, device=dev) * 0.5).bfloat16(); w = (torch.randn(2048, 1024, device=dev) / 32).bfloat16()\n"
        "b = torch.randn(2048, device=dev); dy = (torch.randn(3152, 2048, device=dev) * 0.5).bfloat16()\n"
        "y = Fk.gemm_nt(x, w, b); dx = Fk.gemm_nn(dy, w); dw = Fk.linear_wgrad(x, dy)\n"
        "r = lambda a, b: float((a.float() - b).abs().max() / b.abs().max())\n"
        "print('ERR', r(y, x.float() @ w.float().t() + b), r(dx, dy.float() @ w.float()), r(dw, dy.float().t() @ x.float()))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({}, {"KSMI_GEMM2_OFF": "1"}):
        env = dict(os.environ, PYTHONPATH=root, **extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        errs = [float(v) for v in out.stdout.split("ERR")[1].split()]
        assert errs[0] < 1e-2 and errs[1] < 1e-2 and errs[2] < 2e-3, (extra, errs)


def test_linear_wgrad_every_tile_and_split():
    """gemm2_tn_kernel<MT>: every B-side tile (128 / 96 / 64 columns; the 96-column image has its own LDS layout) in direct mode (one
    split, row-major fp32 gradient) and over split slabs + the reducer, on ragged K / N / rows.  The probe variables KSMI_TN_BT /
    KSMI_TN_SPLIT pin the chooser (read once per process), and the ring depth KSMI_TN_NS its other instances."""
    import os
    import subprocess
    import sys
    code = (
        "import torch\n"
        "from kurosiwo_amd import functional as Fk\n"
        "dev = torch.device('cuda:0')\n"
        "errs = []\n"
        "for rows, K, N in [(3152, 1024, 3072), (1000, 200, 328), (777, 136, 104), (4096, 72, 1000)]:\n"
        "    torch.manual_seed(rows)\n"
        "    x = (torch.randn(rows, K, device=dev) * 0.5).bfloat16(); dy = (torch.randn(rows, N, device=dev) * 0.5).bfloat16()\n"
        "    ref = dy.float().t() @ x.float()\n"
        "    errs.append(float((Fk.linear_wgrad(x, dy) - ref).abs().max() / ref.abs().max()))\n"
        "print('ERR', *errs)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for bt in ("128", "96", "64"):
        # (spl = "1": the two-wave-group instance of round 4 -- each group walks half of the workgroup's reduction range and the halves
        # meet in LDS -- forced wherever it fits, in direct mode and over split slabs; spl = "0": the one-group instances)
        for split, ns, spl in (("1", "3", "0"), ("3", "3", "0"), ("1", "4", "0"), ("2", "5", "0"), ("1", "3", "1"), ("3", "3", "1")):
            env = dict(os.environ, PYTHONPATH=root, KSMI_TN_BT=bt, KSMI_TN_SPLIT=split, KSMI_TN_NS=ns, KSMI_TN_SPL=spl)
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            assert out.returncode == 0, out.stderr[-2000:]
            errs = [float(v) for v in out.stdout.split("ERR")[1].split()]
            assert len(errs) == 4 and max(errs) < 1e-5, (bt, split, ns, spl, errs)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mae_token_shuffles_colsum_mse(dev, dtype):
    """ksmi_gather_rows / scatter_rows / batch_sum / colsum / mse_loss through the C-ABI against torch indexing (models/mae.py:73-122)."""
    import ctypes as C
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import DT, stream_ptr
    lib = _lib.load()
    B, N, nm, Cc = 3, 20, 15, 64
    nv = N - nm
    torch.manual_seed(11)
    idx = torch.rand(B, N, device=dev).argsort(-1)
    src = torch.randn(B, N, Cc, device=dev).to(dtype)
    table = torch.randn(N + 1, Cc, device=dev)
    br = torch.arange(B, device=dev)[:, None]
    vis, msk = idx[:, nm:], idx[:, :nm]
    out = torch.empty(B, nv, Cc, device=dev, dtype=dtype)
    _lib.check(lib.ksmi_gather_rows(src.data_ptr(), idx.data_ptr() + 8 * nm, N, out.data_ptr(), table.data_ptr(), 1, B, N, nv, Cc, DT[dtype], stream_ptr()), "g")
    ref = (src.float()[br, vis] + table[1 + vis]).to(dtype)
    assert torch.equal(out, ref)
    dst = torch.zeros(B, N, Cc, device=dev, dtype=dtype)
    fill = torch.randn(Cc, device=dev)
    _lib.check(lib.ksmi_scatter_rows(out.data_ptr(), None, idx.data_ptr() + 8 * nm, N, table.data_ptr(), dst.data_ptr(), B, nv, N, Cc, DT[dtype], stream_ptr()), "s1")
    _lib.check(lib.ksmi_scatter_rows(None, fill.data_ptr(), idx.data_ptr(), N, table.data_ptr(), dst.data_ptr(), B, nm, N, Cc, DT[dtype], stream_ptr()), "s2")
    want = torch.zeros(B, N, Cc, device=dev)
    want[br, vis] = out.float() + table[vis]
    want[br, msk] = fill.expand(B, nm, Cc) + table[msk]
    assert torch.equal(dst, want.to(dtype))
    bs = torch.empty(N * Cc, device=dev)
    _lib.check(lib.ksmi_batch_sum(dst.data_ptr(), bs.data_ptr(), B, N * Cc, 0, DT[dtype], stream_ptr()), "b")
    assert torch.allclose(bs, dst.float().sum(0).reshape(-1), rtol=1e-5, atol=1e-5)
    cs = torch.ones(Cc, device=dev)
    _lib.check(lib.ksmi_colsum(dst.data_ptr(), B * N, Cc, cs.data_ptr(), 1, DT[dtype], stream_ptr()), "c")
    assert torch.allclose(cs, 1 + dst.float().reshape(-1, Cc).sum(0), rtol=1e-4, atol=1e-4)
    pred, tgt = torch.randn(B * nm, Cc, device=dev).to(dtype), torch.randn(B * nm, Cc, device=dev).to(dtype)
    dpred = torch.empty_like(pred)
    loss = torch.zeros(1, device=dev)
    up = torch.full((1,), 0.25, device=dev)
    ws = torch.zeros(lib.ksmi_mse_workspace() // 4, device=dev)
    _lib.check(lib.ksmi_mse_loss(pred.data_ptr(), tgt.data_ptr(), dpred.data_ptr(), 1.0, up.data_ptr(), loss.data_ptr(), ws.data_ptr(), pred.numel(), DT[dtype], stream_ptr()), "m")
    d = pred.float() - tgt.float()
    assert abs(float(loss) - float((d * d).mean())) < 1e-5 * float((d * d).mean()) + 1e-7
    assert torch.allclose(dpred.float(), (2 * d * 0.25 / d.numel()).to(dtype).float(), rtol=1e-2, atol=1e-8)


@pytest.mark.parametrize("rows,K,N", [(200704, 64, 64), (50176, 128, 512), (12544, 320, 320), (50176, 512, 128), (20000, 256, 64)])
def test_linear_wgrad_split_mode_emits_bias_rows(dev, rows, K, N):
    """Round 5: the split (slab) mode of the token weight gradient also writes the partial column sums of d out, one row per split
    (ksmi_conv_wgrad_fuses_bias == 2; gemm2_tn_kernel<.., BIASA>): the ChangeFormer linears with 12 k - 200 k rows get their bias
    gradient without a channel_sum pass over d out."""
    from kurosiwo_amd import functional as Fk
    g = torch.Generator().manual_seed(rows + K + N)
    x = (torch.randn(rows, K, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    dy = (torch.randn(rows, N, generator=g) * 0.5 + 0.05).to(dev).to(torch.bfloat16)
    dw, db, mode = Fk.linear_wgrad(x, dy, with_bias=True)
    assert mode == 2, mode
    ref_w = dy.float().t() @ x.float()
    ref_b = dy.float().sum(0)
    assert (dw - ref_w).abs().max() < 2e-3 * ref_w.abs().max()
    assert torch.isfinite(db).all()
    assert (db - ref_b).abs().max() < 2e-3 * max(1.0, float(ref_b.abs().max()))
