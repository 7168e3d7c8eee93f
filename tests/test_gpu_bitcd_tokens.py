"""Kernels of the BIT-CD token path (kurosiwo_amd/csrc/bitcd.hip) against the reference's own expressions
(/root/reference/models/bit_cd.py:857-865 tokenizer, :476-524 Cross_Attention behind PreNorm2 / Residual2) evaluated with torch
autograd in float64 on the CPU.  fp32 kernels: 1e-5 relative; bf16 pixels: the rounding of the 32-channel rows."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from kurosiwo_amd import _lib
    return _lib, _lib.load()


def _st(*v):
    return (C.c_int64 * 4)(*v)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_strided_batched_product_and_row_softmax():
    L_, lib = _lib()
    g = torch.Generator().manual_seed(0)
    # per-head attention scores of the token encoder: dots[b][h][i][j] = sum_d q[b][i][h*D+d] k[b][j][h*D+d], q / k inside one qkv buffer
    B, n, H, D = 3, 8, 8, 64
    qkv = torch.randn((B, n, 3 * H * D), generator=g).cuda()
    dots = torch.full((B, H, n, n), 7.0).cuda()
    q, k = qkv[..., :H * D], qkv[..., H * D:2 * H * D]
    rs = 3 * H * D
    L_.check(lib.ksmi_bmm_f32(q.data_ptr(), k.data_ptr(), None, dots.data_ptr(), B, H, n, n, D, _st(n * rs, D, rs, 1), _st(n * rs, D, 1, rs),
                              _st(H * n * n, n * n, n, 1), 0.5, 1, None))
    want = 7.0 + 0.5 * torch.einsum("bihd,bjhd->bhij", q.view(B, n, H, D).double(), k.view(B, n, H, D).double())
    assert _rel(dots, want) < 1e-5
    # a Linear layer with bias, transposed weight by strides, broadcast over the batch
    x, w, bias = torch.randn((40, 32), generator=g).cuda(), torch.randn((64, 32), generator=g).cuda(), torch.randn(64, generator=g).cuda()
    y = torch.empty((40, 64)).cuda()
    L_.check(lib.ksmi_bmm_f32(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), 1, 1, 40, 64, 32, _st(0, 0, 32, 1), _st(0, 0, 1, 32),
                              _st(0, 0, 64, 1), 1.0, 0, None))
    assert _rel(y, torch.nn.functional.linear(x.double(), w.double(), bias.double())) < 1e-5
    s = torch.randn((B * H * n, n), generator=g).cuda()
    p, dp, ds = torch.empty_like(s), torch.randn((B * H * n, n), generator=g).cuda(), torch.empty_like(s)
    L_.check(lib.ksmi_softmax_rows_f32(s.data_ptr(), p.data_ptr(), s.shape[0], n, 0.3, None))
    sd = s.double().cpu().requires_grad_(True)
    pd = (sd * 0.3).softmax(-1)
    pd.backward(dp.double().cpu())
    L_.check(lib.ksmi_softmax_rows_backward_f32(p.data_ptr(), dp.data_ptr(), ds.data_ptr(), s.shape[0], n, 0.3, None))
    assert _rel(p, pd.detach()) < 1e-5 and _rel(ds, sd.grad) < 1e-5
    assert lib.ksmi_softmax_rows_f32(s.data_ptr(), p.data_ptr(), 4, 65, 1.0, None) != 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_semantic_tokens_vs_reference_expression(dtype):
    L_, lib = _lib()
    from kurosiwo_amd.runtime import DT
    g = torch.Generator().manual_seed(1)
    B, N, Cc, L = 3, 23 * 19, 32, 4                     # a pixel count that is not a multiple of anything in the kernel
    x = (torch.randn((2 * B, N, Cc), generator=g) * 1.5).to(dtype).cuda()
    wa = (torch.randn((L, Cc), generator=g) * 0.3).cuda()
    pos = torch.randn((2 * L, Cc), generator=g).cuda()
    tokens, stats = torch.empty((B, 2 * L, Cc)).cuda(), torch.empty((2 * B, L, 2)).cuda()
    L_.check(lib.ksmi_semantic_tokens_forward(x.data_ptr(), wa.data_ptr(), pos.data_ptr(), tokens.data_ptr(), stats.data_ptr(), B, 2, N, Cc, L,
                                              DT[dtype], None))
    xd = x.double().cpu().requires_grad_(True)
    wd = wa.double().cpu().requires_grad_(True)
    sa = torch.softmax(torch.einsum("inc,lc->iln", xd, wd), dim=-1)                      # bit_cd.py:859-862
    tk = torch.einsum("iln,inc->ilc", sa, xd)                                            # :864
    want = torch.cat([tk[:B], tk[B:]], dim=1) + pos.double().cpu()                       # :917 + :881
    assert _rel(tokens, want.detach()) < (1e-5 if dtype == torch.float32 else 1e-6)      # (inputs are the same rounded values)
    dtok = torch.randn((B, 2 * L, Cc), generator=g).cuda()
    want.backward(dtok.double().cpu())
    for accumulate in (0, 1):
        dx = (torch.randn((2 * B, N, Cc), generator=g)).to(dtype).cuda()
        dx0 = dx.clone()
        part = torch.empty((2 * B, L * Cc)).cuda()
        L_.check(lib.ksmi_semantic_tokens_backward(x.data_ptr(), wa.data_ptr(), stats.data_ptr(), dtok.data_ptr(), dx.data_ptr(), part.data_ptr(),
                                                   B, 2, N, Cc, L, accumulate, DT[dtype], None))
        wantdx = xd.grad + (dx0.double().cpu() if accumulate else 0)
        assert _rel(dx, wantdx) < (1e-5 if dtype == torch.float32 else 6e-3)
        assert _rel(part.sum(0).view(L, Cc), wd.grad) < 1e-5


def _cross_reference(x, m, P, heads, scale):
    """Residual2(PreNorm2(Cross_Attention)) (bit_cd.py:436-459, 476-524) in float64: x [img, N, 32] pixels, m [img, L, 32] tokens"""
    ln = lambda t: torch.nn.functional.layer_norm(t, (32,), P["g"], P["b"], 1e-5)
    xn, mn = ln(x), ln(m)
    hd = lambda t: t.view(t.shape[0], t.shape[1], heads, -1).permute(0, 2, 1, 3)
    q, k, v = hd(xn @ P["wq"].T), hd(mn @ P["wk"].T), hd(mn @ P["wv"].T)
    attn = (torch.einsum("bhid,bhjd->bhij", q, k) * scale).softmax(-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v).permute(0, 2, 1, 3).reshape(x.shape[0], x.shape[1], -1)
    return out @ P["wo"].T + P["bo"] + x


@pytest.mark.parametrize("dtype,D", [(torch.float32, 64), (torch.float32, 8), (torch.bfloat16, 64)])
def test_folded_cross_attention_vs_reference_expression(dtype, D):
    L_, lib = _lib()
    from kurosiwo_amd.runtime import DT
    g = torch.Generator().manual_seed(2)
    B, N, Cc, L, H = 2, 300, 32, 4, 8
    inner, scale = H * D, 32 ** -0.5
    x = (torch.randn((2 * B, N, Cc), generator=g) * 1.3 + 0.2).to(dtype)
    tok = torch.randn((B, 2 * L, Cc), generator=g)                                       # encoder output: [b][date*L + l]
    P = {k: torch.randn(shp, generator=g) * sc for k, shp, sc in (("g", (32,), 0.3), ("b", (32,), 0.3), ("wq", (inner, 32), 0.2), ("wk", (inner, 32), 0.2),
                                                                  ("wv", (inner, 32), 0.2), ("wo", (32, inner), 0.1), ("bo", (32,), 0.1))}
    P["g"] = P["g"] + 1.0
    Pd = {k: v.double().requires_grad_(True) for k, v in P.items()}
    xd = x.double().requires_grad_(True)
    tokd = tok.double().requires_grad_(True)
    m = torch.cat([tokd[:, :L], tokd[:, L:]], dim=0)                                     # image order = date-major, as the pixels
    y_ref = _cross_reference(xd, m, Pd, H, scale)
    dy = torch.randn((2 * B, N, Cc), generator=g).to(dtype)
    y_ref.backward(dy.double())
    # token side on the host in float64 (the plan does this with ksmi_bmm_f32): k, v of every token, folded with to_q / to_out
    mn = torch.nn.functional.layer_norm(tok.double(), (32,), P["g"].double(), P["b"].double(), 1e-5)
    k = (mn @ P["wk"].double().T).view(B, 2 * L, H, D)
    v = (mn @ P["wv"].double().T).view(B, 2 * L, H, D)
    A = torch.einsum("brhd,hdc->brhc", k, P["wq"].double().view(H, D, 32)).float().contiguous().cuda()       # [b][row][hd][c]
    Bv = torch.einsum("brhd,chd->brhc", v, P["wo"].double().view(32, H, D)).float().contiguous().cuda()
    xg, y = x.cuda(), torch.empty_like(x).cuda()
    gam, bet, bo = P["g"].cuda(), P["b"].cuda(), P["bo"].cuda()
    L_.check(lib.ksmi_token_cross_forward(xg.data_ptr(), gam.data_ptr(), bet.data_ptr(), A.data_ptr(), Bv.data_ptr(), bo.data_ptr(), y.data_ptr(),
                                          B, 2, N, Cc, H, L, scale, DT[dtype], None))
    f32 = dtype == torch.float32
    assert _rel(y, y_ref.detach()) < (2e-5 if f32 else 8e-3)
    gbuf = dy.clone().cuda()
    dA, dBv = torch.empty_like(A), torch.empty_like(Bv)
    dgam, dbet, dbo = (torch.full((32,), 5.0).cuda() for _ in range(3))
    ws = torch.empty(lib.ksmi_token_cross_bwd_workspace(B, 2, N), dtype=torch.uint8).cuda()
    L_.check(lib.ksmi_token_cross_backward(xg.data_ptr(), gam.data_ptr(), bet.data_ptr(), A.data_ptr(), Bv.data_ptr(), gbuf.data_ptr(), dA.data_ptr(),
                                           dBv.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), dbo.data_ptr(), 0, 1, ws.data_ptr(), B, 2, N, Cc, H, L,
                                           scale, DT[dtype], None))
    assert _rel(gbuf, xd.grad) < (2e-5 if f32 else 8e-3)
    assert _rel(dbo - 5.0, Pd["bo"].grad) < (2e-5 if f32 else 2e-5)
    # dA, dBv back through the fold: gradients of to_q / to_out weights and of k, v
    dAd, dBd = dA.double().cpu(), dBv.double().cpu()
    dwq = torch.einsum("brhc,brhd->hdc", dAd, k).reshape(inner, 32)
    dwo = torch.einsum("brhc,brhd->chd", dBd, v).reshape(32, inner)
    tol = 3e-5 if f32 else 3e-5
    assert _rel(dwq, Pd["wq"].grad) < tol and _rel(dwo, Pd["wo"].grad) < tol
    # LayerNorm parameters: the pixel part from the kernel + the token part through k, v (host float64)
    dk = torch.einsum("brhc,hdc->brhd", dAd, P["wq"].double().view(H, D, 32)).reshape(B, 2 * L, inner)
    dv = torch.einsum("brhc,chd->brhd", dBd, P["wo"].double().view(32, H, D)).reshape(B, 2 * L, inner)
    tokd2 = tok.double().requires_grad_(True)
    g2, b2 = P["g"].double().requires_grad_(True), P["b"].double().requires_grad_(True)
    wk2, wv2 = P["wk"].double().requires_grad_(True), P["wv"].double().requires_grad_(True)
    mn2 = torch.nn.functional.layer_norm(tokd2, (32,), g2, b2, 1e-5)
    ((mn2 @ wk2.T) * dk).sum().backward(retain_graph=True)
    ((mn2 @ wv2.T) * dv).sum().backward()
    assert _rel(dgam.double().cpu() + g2.grad, Pd["g"].grad) < tol and _rel(dbet.double().cpu() + b2.grad, Pd["b"].grad) < tol
    assert _rel(wk2.grad, Pd["wk"].grad) < tol and _rel(tokd2.grad, tokd.grad) < tol
