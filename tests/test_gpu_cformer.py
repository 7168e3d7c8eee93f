"""GPU op-level parity of the ChangeFormer glue kernels (kurosiwo_amd/csrc/cformer.hip) and the conv-epilogue extras
against plain PyTorch fp32 ops on the same device."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 2e-2


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,k,s,p,hw", [(64, 128, 7, 2, 3, 56), (320, 512, 7, 2, 3, 14), (64, 64, 8, 8, 0, 56), (128, 128, 4, 4, 0, 28)])
def test_im2col_linear_equals_conv_and_col2im_is_adjoint(dtype, cin, cout, k, s, p, hw):
    from kurosiwo_amd import functional as KF
    torch.manual_seed(0)
    B = 2
    x = torch.randn(B, cin, hw, hw, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") * (cin * k * k) ** -0.5
    b = torch.randn(cout, device="cuda")
    xq = nhwc(x, dtype)
    col = KF.im2col(xq, k, k, s, p)
    Bc, Ho, Wo, Kp = col.shape
    assert Kp == cin * k * k                                     # whole chunks for every ChangeFormer layer
    y = KF.linear(col.reshape(-1, Kp), w.reshape(cout, -1).contiguous(), b)
    ref = F.conv2d(xq.float().permute(0, 3, 1, 2), w if dtype == torch.float32 else w.bfloat16().float(), b, stride=s, padding=p)
    assert rel(y.reshape(B, Ho, Wo, cout).permute(0, 3, 1, 2), ref) < tol(dtype)
    # adjoint: <im2col(x), g> == <x, col2im(g)>
    g = torch.randn_like(col.float()).to(dtype)
    dx = KF.col2im(g, cin, hw, hw, k, k, s, p)
    lhs = float((col.double() * g.double()).sum())
    rhs = float((xq.double() * dx.double()).sum())
    assert abs(lhs - rhs) <= (1e-4 if dtype == torch.float32 else 3e-2) * abs(lhs) + 1e-3
    # weight gradient through the uniform k-chunk walk (K = cin*k*k up to 15680 = 490 bf16 chunks)
    dy = torch.randn(B * Ho * Wo, cout, device="cuda").to(dtype)
    dw = KF.linear_wgrad(col.reshape(-1, Kp), dy)
    ref_dw = dy.float().t() @ col.reshape(-1, Kp).float()
    assert rel(dw, ref_dw) < (1e-4 if dtype == torch.float32 else 1e-2)
    dcol = KF.linear_dgrad(dy, w.reshape(cout, -1).contiguous())
    ref_dcol = dy.float() @ (w.reshape(cout, -1) if dtype == torch.float32 else w.reshape(cout, -1).bfloat16().float())
    assert rel(dcol, ref_dcol) < tol(dtype)


def test_patch_embed_from_nchw_image():
    from kurosiwo_amd import functional as KF
    torch.manual_seed(1)
    x = torch.randn(2, 2, 224, 224, device="cuda")
    w = torch.randn(64, 2, 7, 7, device="cuda") * 0.1
    col = KF.im2col(x, 7, 7, 4, 3, nchw_image=True, dtype=torch.float32)
    assert col.shape == (2, 56, 56, 112)
    wp = torch.zeros(64, 112, device="cuda")
    wp[:, :98] = w.reshape(64, 98)
    y = KF.linear(col.reshape(-1, 112), wp, None)
    ref = F.conv2d(x, w, None, stride=4, padding=3)
    assert rel(y.reshape(2, 56, 56, 64).permute(0, 3, 1, 2), ref) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Cc,H,W", [(2, 256, 28, 28), (3, 64, 7, 7), (2, 40, 10, 33), (2, 256, 56, 56), (1, 2048, 7, 7), (2, 1280, 14, 14), (1, 8, 5, 1)])
def test_dwconv_gelu_forward_backward(dtype, B, Cc, H, W):
    """the row-walking kernels (cformer.hip dwconv3x3_row_kernel / dwconv3x3_wgrad_row_kernel): the four maps of the MiT encoder plus
    ragged widths (segment tails, a single column, more channel vectors than one block column)"""
    from kurosiwo_amd import functional as KF
    torch.manual_seed(2)
    x = torch.randn(B, Cc, H, W, device="cuda")
    w = (torch.randn(Cc, 1, 3, 3, device="cuda") * 0.4).requires_grad_(True)
    b = torch.randn(Cc, device="cuda").requires_grad_(True)
    xq = nhwc(x, dtype)
    xr = xq.float().permute(0, 3, 1, 2).requires_grad_(True)
    zr = F.conv2d(xr, w, b, padding=1, groups=Cc)
    gr = F.gelu(zr)
    z, g = KF.dwconv3x3_gelu(xq, w.detach(), b.detach())
    assert rel(z.permute(0, 3, 1, 2), zr) < tol(dtype) and rel(g.permute(0, 3, 1, 2), gr) < tol(dtype)
    dz = torch.randn(B, Cc, H, W, device="cuda")
    dzq = nhwc(dz, dtype)
    zr.backward(dzq.float().permute(0, 3, 1, 2))
    dx, dw, db = KF.dwconv3x3_backward(xq, dzq, w.detach())
    assert rel(dx.permute(0, 3, 1, 2), xr.grad) < tol(dtype)
    assert rel(dw, w.grad) < 1e-4 and rel(db, b.grad) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Cc,heads,Nq", [(64, 1, 3136), (128, 2, 784), (320, 4, 196), (512, 8, 49)])
def test_sr_attention_forward_backward(dtype, Cc, heads, Nq):
    from kurosiwo_amd import functional as KF
    torch.manual_seed(3)
    B, Nk, d = 2, 49, Cc // heads
    q = torch.randn(B * Nq, Cc, device="cuda").to(dtype)
    kv = torch.randn(B * Nk, 2 * Cc, device="cuda").to(dtype)
    qr = q.float().requires_grad_(True)
    kvr = kv.float().requires_grad_(True)
    qq = qr.reshape(B, Nq, heads, d).permute(0, 2, 1, 3)
    kk = kvr.reshape(B, Nk, 2, heads, d).permute(2, 0, 3, 1, 4)
    attn = ((qq @ kk[0].transpose(-2, -1)) * d ** -0.5).softmax(-1)
    ref = (attn @ kk[1]).transpose(1, 2).reshape(B * Nq, Cc)
    out = KF.sr_attention(q, kv, B, Nq, Nk, heads)
    assert rel(out, ref) < tol(dtype)
    do = torch.randn(B * Nq, Cc, device="cuda").to(dtype)
    ref.backward(do.float())
    dq, dkv = KF.sr_attention_backward(q, kv, out, do, B, Nq, Nk, heads)
    assert rel(dq, qr.grad) < tol(dtype) and rel(dkv, kvr.grad) < tol(dtype)


def _rng_state(seed, step):
    return torch.tensor([seed, step], dtype=torch.int32, device="cuda")


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout_apply_regenerates_the_oracle_masks(dtype):
    """ksmi_dropout_apply (nn.Dropout :107,162 + DropPath :236-241 + residual add): the masks are bit-identical to oracle/rng_ref.py,
    and ksmi_rng_advance moves the stream to the next step."""
    from kurosiwo_amd import _lib, functional as KF
    from kurosiwo_amd.runtime import stream_ptr
    from oracle import rng_ref as G
    torch.manual_seed(11)
    B2, N, Cc = 4, 49, 64
    rows = B2 * N
    seed, step, site, psite = 99, 5, 8 * 7 + G.SITE_MLP2, 8 * 7 + G.SITE_PATH_MLP
    st = _rng_state(seed, step)
    x = (torch.randn(rows, Cc, device="cuda") + 3.0).to(dtype)           # no zeros: the mask is readable from the output
    r = torch.randn(rows, Cc, device="cuda").to(dtype)
    m_el = torch.from_numpy(G.scale_mask(seed, step, site, 0.1, 0, (rows, Cc))).cuda()
    m_path = torch.from_numpy(G.scale_mask(seed, step, psite, 0.3, 0, (B2,))).cuda().repeat_interleave(N)[:, None]
    y = KF.dropout_apply(x, st, p=0.1, site=site)
    assert torch.equal(y != 0, m_el > 0)
    assert rel(y, x.float() * m_el) < tol(dtype)
    y = KF.dropout_apply(x, st, path_p=0.3, path_site=psite, rows_per_sample=N)
    assert torch.equal(y != 0, (m_path > 0).expand(rows, Cc))
    y = KF.dropout_apply(x, st, p=0.1, site=site, path_p=0.3, path_site=psite, rows_per_sample=N, resid=r)
    assert rel(y, r.float() + x.float() * m_el * m_path) < tol(dtype)
    y2 = KF.dropout_apply(x.clone(), st, p=0.1, site=site, out=None)
    xin = x.clone()
    KF.dropout_apply(xin, st, p=0.1, site=site, out=xin)                 # in place (Mlp.drop on the activation)
    assert torch.equal(xin, y2)
    _lib.check(_lib.load().ksmi_rng_advance(st.data_ptr(), stream_ptr()), "rng_advance")
    assert st.cpu().tolist() == [seed, step + 1]
    y3 = KF.dropout_apply(x, st, p=0.1, site=site)
    assert torch.equal(y3 != 0, torch.from_numpy(G.scale_mask(seed, step + 1, site, 0.1, 0, (rows, Cc))).cuda() > 0)
    with pytest.raises(_lib.KsmiError):
        KF.dropout_apply(x[:, :Cc - 1].contiguous(), st, p=0.1, site=site)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Cc,heads,Nq", [(64, 1, 3136), (128, 2, 784), (320, 4, 196), (512, 8, 49)])
def test_sr_attention_with_attention_dropout(dtype, Cc, heads, Nq):
    """attn_drop (changeformer.py:160,203) inside the fused attention kernels (MFMA for bf16, VALU for fp32): forward and both backward
    passes regenerate the mask of oracle/rng_ref.py at element ((b*heads + h)*Nq + q)*Nk + key."""
    from kurosiwo_amd import functional as KF
    from oracle import rng_ref as G
    torch.manual_seed(3)
    B, Nk, d = 2, 49, Cc // heads
    seed, step, site, p = 1234, 7, 8 * 3 + G.SITE_ATTN, 0.1
    st = _rng_state(seed, step)
    q = torch.randn(B * Nq, Cc, device="cuda").to(dtype)
    kv = torch.randn(B * Nk, 2 * Cc, device="cuda").to(dtype)
    qr = q.float().requires_grad_(True)
    kvr = kv.float().requires_grad_(True)
    qq = qr.reshape(B, Nq, heads, d).permute(0, 2, 1, 3)
    kk = kvr.reshape(B, Nk, 2, heads, d).permute(2, 0, 3, 1, 4)
    mask = torch.from_numpy(G.scale_mask(seed, step, site, p, 0, (B, heads, Nq, Nk))).cuda()
    attn = ((qq @ kk[0].transpose(-2, -1)) * d ** -0.5).softmax(-1) * mask
    ref = (attn @ kk[1]).transpose(1, 2).reshape(B * Nq, Cc)
    out = KF.sr_attention(q, kv, B, Nq, Nk, heads, p=p, site=site, rng_state=st)
    assert rel(out, ref) < tol(dtype)
    plain = KF.sr_attention(q, kv, B, Nq, Nk, heads)
    assert rel(plain, ref) > 10 * tol(dtype)                              # the mask is really applied
    do = torch.randn(B * Nq, Cc, device="cuda").to(dtype)
    ref.backward(do.float())
    dq, dkv = KF.sr_attention_backward(q, kv, out, do, B, Nq, Nk, heads, p=p, site=site, rng_state=st)
    assert rel(dq, qr.grad) < tol(dtype) and rel(dkv, kvr.grad) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hi,ho", [(7, 14), (7, 56), (14, 56), (28, 56), (5, 13), (9, 9)])
def test_bilinear_forward_backward(dtype, hi, ho):
    from kurosiwo_amd import functional as KF
    torch.manual_seed(4)
    B, Cc = 2, 64
    x = torch.randn(B, Cc, hi, hi, device="cuda")
    add = torch.randn(B, Cc, ho, ho, device="cuda")
    xq, aq = nhwc(x, dtype), nhwc(add, dtype)
    xr = xq.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.interpolate(xr, size=(ho, ho), mode="bilinear", align_corners=False)
    assert rel(KF.bilinear(xq, ho, ho).permute(0, 3, 1, 2), ref) < tol(dtype)
    assert rel(KF.bilinear(xq, ho, ho, add=aq).permute(0, 3, 1, 2), ref + aq.float().permute(0, 3, 1, 2)) < tol(dtype)
    dy = nhwc(torch.randn(B, Cc, ho, ho, device="cuda"), dtype)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    dx = KF.bilinear_backward(dy, hi, hi)
    assert rel(dx.permute(0, 3, 1, 2), xr.grad) < tol(dtype)
    dx2 = KF.bilinear_backward(dy, hi, hi, out=dx.clone())
    assert rel(dx2.permute(0, 3, 1, 2), 2 * xr.grad) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_epilogue_relu_stats_and_scaled_residual(dtype):
    from kurosiwo_amd import functional as KF
    torch.manual_seed(5)
    B, Cc, H, W = 2, 64, 28, 28
    x = nhwc(torch.randn(B, Cc, H, W, device="cuda"), dtype)
    w = torch.randn(Cc, Cc, 3, 3, device="cuda") * (Cc * 9) ** -0.5
    b = torch.randn(Cc, device="cuda") * 0.1
    wq = w if dtype == torch.float32 else w.bfloat16().float()
    conv = F.conv2d(x.float().permute(0, 3, 1, 2), wq, b, padding=1)
    out, stats = KF.conv3x3([x], w, b, want_stats=True, relu_out=1)
    ref = F.relu(conv)
    assert rel(out.permute(0, 3, 1, 2), ref) < tol(dtype)
    s = stats.sum(0)                                              # [2, Npad]: sum, sum of squares of the ReLU output
    assert rel(s[0, :Cc], ref.sum((0, 2, 3))) < (1e-4 if dtype == torch.float32 else 1e-2)
    assert rel(s[1, :Cc], (ref * ref).sum((0, 2, 3))) < (1e-4 if dtype == torch.float32 else 1e-2)
    out2, _ = KF.conv3x3([x], w, b, alpha=0.1, resid=x)
    assert rel(out2.permute(0, 3, 1, 2), conv * 0.1 + x.float().permute(0, 3, 1, 2)) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("relu_mask", [0, 1])
def test_bn_bwd_apply_and_affine_and_sigmoid_out(dtype, relu_mask):
    import ctypes as C
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import DT, stream_ptr
    lib = _lib.load()
    torch.manual_seed(6)
    B, Cc, H, W = 2, 64, 14, 14
    v = torch.randn(B, Cc, H, W, device="cuda")
    vq = nhwc(v, dtype)
    vr = vq.float().permute(0, 3, 1, 2).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(Cc, device="cuda")).requires_grad_(True)
    beta = (0.1 * torch.randn(Cc, device="cuda")).requires_grad_(True)
    r = F.relu(vr) if relu_mask else vr
    y = F.batch_norm(r, None, None, gamma, beta, True, 0.1, 1e-5)
    dy = nhwc(torch.randn(B, Cc, H, W, device="cuda"), dtype)
    y.backward(dy.float().permute(0, 3, 1, 2))
    rq = nhwc(r.detach(), dtype)
    rf = rq.float()
    mean = rf.mean((0, 1, 2))
    rstd = (rf.var((0, 1, 2), unbiased=False) + 1e-5).rsqrt()
    rhat = (rf - mean) * rstd
    sums = torch.stack([dy.float().sum((0, 1, 2)), (dy.float() * rhat).sum((0, 1, 2))]).contiguous()
    dv = torch.empty_like(rq)
    npix = B * H * W
    _lib.check(lib.ksmi_bn_bwd_apply(dy.data_ptr(), rq.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), sums.data_ptr(),
                                     dv.data_ptr(), relu_mask, float(npix), npix, Cc, DT[dtype], stream_ptr()))
    assert rel(dv.permute(0, 3, 1, 2), vr.grad) < (1e-4 if dtype == torch.float32 else 3e-2)
    # affine: y = 0.5 * relu(x*scale + shift)
    out = torch.empty_like(rq)
    _lib.check(lib.ksmi_affine(rq.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), npix, Cc, 1, C.c_float(0.5), DT[dtype], stream_ptr()))
    assert rel(out, 0.5 * F.relu(rq.float() * gamma.detach() + beta.detach())) < tol(dtype)
    # sigmoid head conversion and its adjoint
    Cs = 8
    xh = torch.randn(B, H * W, Cs, device="cuda").to(dtype)
    yo = torch.empty(B, 3, H * W, device="cuda")
    _lib.check(lib.ksmi_out_to_nchw(xh.data_ptr(), yo.data_ptr(), B, 3, Cs, H * W, 1, DT[dtype], stream_ptr()))
    ref = torch.sigmoid(xh.float()[:, :, :3].permute(0, 2, 1))
    assert rel(yo, ref) < 1e-5
    g = torch.randn_like(yo)
    dxh = torch.empty_like(xh)
    _lib.check(lib.ksmi_dout_to_nhwc(g.data_ptr(), yo.data_ptr(), dxh.data_ptr(), B, 3, Cs, H * W, 1, DT[dtype], stream_ptr()))
    refd = (g * ref * (1 - ref)).permute(0, 2, 1)
    assert rel(dxh.float()[:, :, :3], refd) < tol(dtype) and float(dxh.float()[:, :, 3:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,k,s,p,hw", [(64, 128, 7, 2, 3, 56), (320, 512, 7, 2, 3, 14), (64, 64, 8, 8, 0, 56), (128, 128, 4, 4, 0, 28)])
def test_channel_fastest_im2col_family(dtype, cin, cout, k, s, p, hw):
    """ksmi_im2col_tc / col2im_tc / weight_to_tc / grad_from_tc (K index = tap * Cin + c) against the OIHW-ordered family: the
    column matrices are permutations of each other, the re-ordered weight gives the same product, and the weight gradient comes
    back in OIHW order."""
    from kurosiwo_amd import _lib
    from kurosiwo_amd import functional as KF
    from kurosiwo_amd.runtime import DT, stream_ptr
    lib = _lib.load()
    torch.manual_seed(1)
    B, T = 2, k * k
    K = cin * T
    x = nhwc(torch.randn(B, cin, hw, hw, device="cuda"), dtype)
    col = KF.im2col(x, k, k, s, p)
    _, Ho, Wo, Kp = col.shape
    assert Kp == K
    col_tc = torch.empty_like(col)
    _lib.check(lib.ksmi_im2col_tc(x.data_ptr(), col_tc.data_ptr(), B, cin, hw, hw, Ho, Wo, k, k, s, p, K, DT[dtype], stream_ptr()), "im2col_tc")
    assert torch.equal(col_tc.reshape(-1, T, cin), col.reshape(-1, cin, T).transpose(1, 2))
    g = torch.randn(B, Ho, Wo, K, device="cuda").to(dtype)
    g_tc = g.reshape(-1, cin, T).transpose(1, 2).contiguous().reshape(B, Ho, Wo, K)
    dx = KF.col2im(g, cin, hw, hw, k, k, s, p)
    dx_tc = torch.full_like(dx, 0.5)
    _lib.check(lib.ksmi_col2im_tc(g_tc.data_ptr(), dx_tc.data_ptr(), 1, B, cin, hw, hw, Ho, Wo, k, k, s, p, K, DT[dtype], stream_ptr()), "col2im_tc")
    assert rel(dx_tc.float() - 0.5, dx) < (1e-5 if dtype == torch.float32 else 2e-2)
    w = torch.randn(cout, cin, k, k, device="cuda")
    w_tc = torch.empty(cout, K, device="cuda", dtype=dtype)
    _lib.check(lib.ksmi_weight_to_tc(w.data_ptr(), w_tc.data_ptr(), cout, cin, T, K, DT[dtype], stream_ptr()), "weight_to_tc")
    assert torch.equal(w_tc, w.reshape(cout, cin, T).transpose(1, 2).reshape(cout, K).to(dtype))
    gt = torch.randn(cout, K, device="cuda")
    grad = torch.ones(cout, cin, k, k, device="cuda")
    _lib.check(lib.ksmi_grad_from_tc(gt.data_ptr(), grad.data_ptr(), cout, cin, T, K, 1, stream_ptr()), "grad_from_tc")
    assert torch.equal(grad, 1 + gt.reshape(cout, T, cin).transpose(1, 2).reshape(cout, cin, k, k))


def test_fused_mlp_dropout_is_bit_identical_to_the_separate_passes(monkeypatch):
    """Round 6: Mlp.drop (models/changeformer.py:130) rides on the depth-wise kernel's store of the activation and on the gelu' pass of the
    backward (ksmi_dwconv3x3_gelu_forward_drop, ksmi_gelu_backward_drop) instead of two passes of ksmi_dropout_apply: same draws (site,
    flat element index), products formed on the rounded values -- three train steps equal bit for bit, 26 launches fewer per step."""
    import torch
    from kurosiwo_amd.changeformer import ChangeFormerV6
    from kurosiwo_amd.trainer import CDTrainStep
    g = torch.Generator().manual_seed(3)
    data = [(torch.randn(2, 2, 224, 224, generator=g), torch.randn(2, 2, 224, 224, generator=g), torch.randint(0, 3, (2, 224, 224), generator=g)) for _ in range(3)]
    out = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("KSMI_CF_FUSE_DROP", fuse)
        torch.manual_seed(5)
        m = ChangeFormerV6(input_nc=2, output_nc=3, decoder_softmax=True, embed_dim=64, precision="bf16").cuda().train()
        m.manual_seed(11, 0)
        st = CDTrainStep(m, 2, 224, 224, "ce+dice", (1.0, 1.0, 1.0), lr=1e-3)
        names = [n for _, _, n, _ in st.plan.fwd.calls + st.plan.bwd.calls]
        losses = [st.step(a.cuda(), b.cuda(), y.cuda()).clone() for a, b, y in data]
        torch.cuda.synchronize()
        out.append((losses, m.flat_params.clone(), m.flat_grads.clone(), names.count("ksmi_dropout_apply"),
                    names.count("ksmi_dwconv3x3_gelu_forward_drop") + names.count("ksmi_gelu_backward_drop")))
    assert out[0][4] == 0 and out[1][4] == 26 and out[0][3] - out[1][3] == 26, (out[0][3:], out[1][3:])
    assert all(torch.equal(a, b) for a, b in zip(out[0][0], out[1][0]))
    assert torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("B", [5, 6])
def test_decoder_phase_convolutions_at_ragged_batch_sizes(B):
    """at per-rank batches of 5 / 6 the 56 x 56 phase convolutions of the k4 s2 transposed convolutions fell into a
    window where ksmi_igemm4_geom returned a tile shape without a compiled 2 x 2 instance and the forward raised instead of falling back
    (ragged last validation batches, small data-parallel shards).  A whole train step and an evaluation forward at those sizes."""
    import torch
    from kurosiwo_amd.changeformer import ChangeFormerV6
    from kurosiwo_amd.trainer import CDTrainStep
    torch.manual_seed(1)
    m = ChangeFormerV6(input_nc=2, output_nc=3, decoder_softmax=True, embed_dim=256, precision="bf16").cuda().train()
    st = CDTrainStep(m, B, 224, 224, "ce+dice", (1.0, 1.0, 1.0), lr=1e-4)
    g = torch.Generator().manual_seed(B)
    xa, xb = torch.randn(B, 2, 224, 224, generator=g), torch.randn(B, 2, 224, 224, generator=g)
    y = torch.randint(0, 3, (B, 224, 224), generator=g)
    loss = st.step(xa.cuda(), xb.cuda(), y.cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all() and torch.isfinite(m.flat_grads).all()
    m.eval()
    with torch.no_grad():
        out = m(xa.cuda(), xb.cuda())
    assert out[-1].shape == (B, 3, 224, 224) and torch.isfinite(out[-1]).all()
