"""Functional (allocate-and-run) wrappers over the C-ABI conv family; used by the tests and
by one-off callers.  The model plans (snunet_plan.py) build the same descriptors once and
replay them instead."""
import ctypes as C

import torch

from . import _lib
from .runtime import DT, SrcSpec, conv_grid_m, conv_npad, conv_stats_rows, make_conv, make_pack, make_wgrad, packed_weight_numel, require_gpu, stream_ptr


def to_nhwc(x_nchw, dtype):
    """NCHW fp32 -> NHWC `dtype` through the HIP layout kernel."""
    require_gpu(x_nchw)
    B, Cc, H, W = x_nchw.shape
    y = torch.empty((B, H, W, Cc), dtype=dtype, device=x_nchw.device)
    _lib.check(_lib.load().ksmi_nchw_to_nhwc(x_nchw.contiguous().float().data_ptr(), y.data_ptr(), B, Cc, H * W, DT[dtype], stream_ptr()))
    return y


def to_nchw(x_nhwc):
    B, H, W, Cc = x_nhwc.shape
    y = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x_nhwc.device)
    _lib.check(_lib.load().ksmi_nhwc_to_nchw(x_nhwc.data_ptr(), y.data_ptr(), B, Cc, H * W, DT[x_nhwc.dtype], stream_ptr()))
    return y


def _pack(weight, table, taps, N, n_mod, sK, sN, sD, sT, flip, dtype):
    Npad = conv_npad(N)
    out = torch.empty(packed_weight_numel(table, taps, Npad, dtype), dtype=dtype, device=weight.device)
    d = make_pack(weight, out, table, taps, N, Npad, n_mod, sK, sN, sD, sT, flip)
    _lib.check(_lib.load().ksmi_pack_weights(C.byref(d), DT[dtype], stream_ptr()), "pack_weights")
    return out


def conv3x3(xs, weight, bias=None, affine=None, want_stats=False, alpha=0.0, relu_out=0, resid=None, mask=None, out=None, accumulate=0,
            gate=None):
    """xs: list of NHWC tensors (virtual concat); weight [N, sum C, 3, 3] fp32; affine=(scale, shift, relu)
    applies to a single source; out = [relu](alpha*(conv + bias) + resid).  mask = (m NHWC, mean, rstd, scale, shift): the
    ReLU-mask + BatchNorm-backward epilogue of ksmi_conv_desc (result zeroed where m*scale+shift <= 0; stats = (sum v, sum v*xhat)).
    out / accumulate: destination tensor and dst += result.  gate = (out_block NHWC, z NHWC, mean, rstd): the gate epilogue of
    ksmi_conv_desc (total = result + old destination, zeroed where out_block <= 0; stats = (sum v, sum v*zhat)).
    Returns (out NHWC, stats[rows,2,Npad] or None)."""
    dtype = xs[0].dtype
    B, H, W, _ = xs[0].shape
    N = weight.shape[0]
    Ktot = sum(x.shape[3] for x in xs)
    srcs = [SrcSpec(x, x.shape[3]) for x in xs]
    if affine is not None:
        srcs[0].scale, srcs[0].shift, srcs[0].relu = affine
    if out is None:
        out = torch.empty((B, H, W, N), dtype=dtype, device=xs[0].device)
    # (a destination whose rows are wider than N -- the 8-channel-stride tensors of the 2- / 3-class heads -- keeps its channel stride)
    d, table = make_conv(srcs, [(out, out.shape[3], 0, 0, N, accumulate)], out, bias, None, B, H, W, H, W, 3, 3, 1, 1, N, dtype,
                         alpha=alpha, relu_out=relu_out, resid=None if resid is None else (resid, resid.shape[3]), mask=mask)
    wpk = _pack(weight.contiguous(), table, 9, N, N, 9, Ktot * 9, 0, 1, 0, dtype)
    d.wpk = wpk.data_ptr()
    if gate is not None:
        d.gate_src, d.xhat_src, d.g_mean, d.g_rstd = (t.data_ptr() for t in gate)
        if not _lib.load().ksmi_conv_gate_supported(C.byref(d), DT[dtype]):
            raise _lib.KsmiError("conv3x3: the gate epilogue is not available for this shape")
    stats = None
    if want_stats:
        stats = torch.zeros((conv_stats_rows(d, dtype), 2, d.Npad), dtype=torch.float32, device=out.device)
        d.stats = stats.data_ptr()
    _lib.check(_lib.load().ksmi_conv_forward(C.byref(d), DT[dtype], stream_ptr()), "conv_forward")
    return out, stats


def conv3x3_dgrad(dy, weight, splits):
    """dX for each concat source (channel counts `splits`) of y = conv3x3(cat(xs), weight)."""
    dtype = dy.dtype
    B, H, W, N = dy.shape
    Ktot = sum(splits)
    outs = [torch.empty((B, H, W, c), dtype=dtype, device=dy.device) for c in splits]
    dsts, nb = [], 0
    for o, c in zip(outs, splits):
        dsts.append((o, c, 0, nb, c, 0))
        nb += c
    d, table = make_conv([SrcSpec(dy, N)], dsts, dy, None, None, B, H, W, H, W, 3, 3, 1, 1, Ktot, dtype)
    wpk = _pack(weight.contiguous(), table, 9, Ktot, Ktot, Ktot * 9, 9, 0, 1, 1, dtype)
    d.wpk = wpk.data_ptr()
    _lib.check(_lib.load().ksmi_conv_forward(C.byref(d), DT[dtype], stream_ptr()), "conv_dgrad")
    return outs


def conv3x3_wgrad(xs, dy, affine=None, n_real=None):
    """n_real: output channels that exist when d out is stored with a wider channel stride (the 8-channel-stride 2- / 3-class heads)."""
    dtype = dy.dtype
    B, H, W, dyC = dy.shape
    N = dyC if n_real is None else n_real
    Ktot = sum(x.shape[3] for x in xs)
    srcs = [SrcSpec(x, x.shape[3]) for x in xs]
    if affine is not None:
        srcs[0].scale, srcs[0].shift, srcs[0].relu = affine
    grad = torch.zeros((N, Ktot, 3, 3), dtype=torch.float32, device=dy.device)
    d, ws = make_wgrad(srcs, dy, dyC, 0, N, grad, 9, Ktot * 9, 1, 0, B, H, W, H, W, 3, 3, 1, 1, dtype)
    wsb = torch.empty(max(ws, 16), dtype=torch.uint8, device=dy.device)
    d.partial = wsb.data_ptr()
    _lib.check(_lib.load().ksmi_conv_wgrad(C.byref(d), DT[dtype], stream_ptr()), "conv_wgrad")
    return grad


def deconv2x2(x, weight, bias):
    """ConvTranspose2d(C, C, 2, stride=2): x NHWC [B,H,W,Cin], weight [Cin, Cout, 2, 2]."""
    dtype = x.dtype
    B, H, W, Cin = x.shape
    Cout = weight.shape[1]
    y = torch.empty((B, 2 * H, 2 * W, Cout), dtype=dtype, device=x.device)
    d, table = make_conv([SrcSpec(x, Cin)], [(y, Cout, 0, 0, 4 * Cout, 0)], x, bias, None, B, H, W, H, W, 1, 1, 1, 0,
                         4 * Cout, dtype, ps_cout=Cout)
    wpk = _pack(weight.contiguous(), table, 1, 4 * Cout, Cout, Cout * 4, 4, 1, 0, 0, dtype)
    d.wpk = wpk.data_ptr()
    _lib.check(_lib.load().ksmi_conv_forward(C.byref(d), DT[dtype], stream_ptr()), "deconv_forward")
    return y


def deconv2x2_backward(x, dy, weight):
    """returns (dx, dW) of y = deconv2x2(x, weight)."""
    dtype = x.dtype
    B, H, W, Cin = x.shape
    Cout = weight.shape[1]
    dx = torch.empty_like(x)
    s2 = [SrcSpec(dy, Cout)]
    d2, t2 = make_conv(s2, [(dx, Cin, 0, 0, Cin, 0)], dy, None, None, B, 2 * H, 2 * W, H, W, 2, 2, 2, 0, Cin, dtype)
    w2 = _pack(weight.contiguous(), t2, 4, Cin, Cin, 4, Cout * 4, 0, 1, 0, dtype)
    d2.wpk = w2.data_ptr()
    _lib.check(_lib.load().ksmi_conv_forward(C.byref(d2), DT[dtype], stream_ptr()), "deconv_dgrad")
    grad = torch.zeros_like(weight)
    dw, ws = make_wgrad(s2, x, Cin, 0, Cin, grad, 4, Cout * 4, 1, 0, B, 2 * H, 2 * W, H, W, 2, 2, 2, 0, dtype)
    wsb = torch.empty(max(ws, 16), dtype=torch.uint8, device=x.device)
    dw.partial = wsb.data_ptr()
    _lib.check(_lib.load().ksmi_conv_wgrad(C.byref(dw), DT[dtype], stream_ptr()), "deconv_wgrad")
    return dx, grad


# ---------------------------------------------------------------------------------------------------
# token ops (FloodViT path): x is [rows, C] in the activation dtype
# ---------------------------------------------------------------------------------------------------
def linear(x, weight, bias=None, out=None, accumulate=0):
    """y[rows, N] (+)= x[rows, Cin] @ weight[N, Cin]^T + bias  — a 1x1 implicit GEMM over a rows x 1 'image'."""
    dtype = x.dtype
    rows, Cin = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((rows, N), dtype=dtype, device=x.device)
    d, table = make_conv([SrcSpec(x, Cin)], [(out, N, 0, 0, N, accumulate)], out, bias, None, 1, rows, 1, rows, 1, 1, 1, 1, 0, N, dtype)
    wpk = _pack(weight.contiguous(), table, 1, N, N, 1, Cin, 0, 0, 0, dtype)
    d.wpk = wpk.data_ptr()
    _lib.check(_lib.load().ksmi_conv_forward(C.byref(d), DT[dtype], stream_ptr()), "linear")
    return out


def linear_dgrad(dy, weight, out=None, accumulate=0):
    """dx[rows, Cin] (+)= dy[rows, N] @ weight[N, Cin]."""
    dtype = dy.dtype
    rows, N = dy.shape
    Cin = weight.shape[1]
    if out is None:
        out = torch.empty((rows, Cin), dtype=dtype, device=dy.device)
    d, table = make_conv([SrcSpec(dy, N)], [(out, Cin, 0, 0, Cin, accumulate)], out, None, None, 1, rows, 1, rows, 1, 1, 1, 1, 0, Cin, dtype)
    wpk = _pack(weight.contiguous(), table, 1, Cin, Cin, Cin, 1, 0, 0, 0, dtype)
    d.wpk = wpk.data_ptr()
    _lib.check(_lib.load().ksmi_conv_forward(C.byref(d), DT[dtype], stream_ptr()), "linear_dgrad")
    return out


def linear_wgrad(x, dy, with_bias=False):
    """dW[N, Cin] = dy^T @ x (fp32).  with_bias: also returns (db[N] or None, mode): the bias gradient when the weight-gradient launch
    computes it (ksmi_conv_wgrad_fuses_bias: 1 = written by the launch, 2 = one partial row per split, summed here), else None."""
    dtype = x.dtype
    rows, Cin = x.shape
    N = dy.shape[1]
    grad = torch.zeros((N, Cin), dtype=torch.float32, device=x.device)
    d, ws = make_wgrad([SrcSpec(x, Cin)], dy, N, 0, N, grad, 1, Cin, 0, 0, 1, rows, 1, rows, 1, 1, 1, 1, 0, dtype)
    wsb = torch.empty(max(ws, 16), dtype=torch.uint8, device=x.device)
    d.partial = wsb.data_ptr()
    lib = _lib.load()
    mode, bias_buf = 0, None
    if with_bias:
        bias_buf = torch.full((max(d.nsplit, 1), N), float("nan"), dtype=torch.float32, device=x.device)
        d.bias_grad = bias_buf.data_ptr()
        mode = int(lib.ksmi_conv_wgrad_fuses_bias(C.byref(d), DT[dtype]))
        if mode == 0:
            d.bias_grad = None
    _lib.check(lib.ksmi_conv_wgrad(C.byref(d), DT[dtype], stream_ptr()), "linear_wgrad")
    if not with_bias:
        return grad
    db = None if mode == 0 else (bias_buf[0].clone() if mode == 1 else bias_buf[:d.nsplit].sum(0))
    return grad, db, mode


def layernorm(x, gamma, beta, eps=1e-5):
    rows, Cc = x.shape
    lib = _lib.load()
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(lib.ksmi_layernorm_forward(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), rows, Cc, eps, DT[x.dtype], stream_ptr()), "layernorm_forward")
    return y, mean, rstd


def layernorm_backward(dy, x, mean, rstd, gamma):
    rows, Cc = x.shape
    lib = _lib.load()
    dx = torch.empty_like(x)
    nblk = lib.ksmi_layernorm_bwd_blocks(rows)
    partial = torch.empty((nblk, 2, Cc), dtype=torch.float32, device=x.device)
    _lib.check(lib.ksmi_layernorm_backward(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                           dx.data_ptr(), 0, partial.data_ptr(), rows, Cc, DT[x.dtype], stream_ptr()), "layernorm_backward")
    dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device)
    _lib.check(lib.ksmi_reduce_rows(partial.data_ptr(), nblk, 2, Cc, Cc, None, dgamma.data_ptr(), dbeta.data_ptr(), 0, stream_ptr()), "reduce_rows")
    return dx, dgamma, dbeta


def gelu(x):
    y = torch.empty_like(x)
    _lib.check(_lib.load().ksmi_gelu_forward(x.data_ptr(), y.data_ptr(), x.numel(), DT[x.dtype], stream_ptr()), "gelu")
    return y


def gelu_backward(dy, x):
    dx = torch.empty_like(x)
    _lib.check(_lib.load().ksmi_gelu_backward(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), DT[x.dtype], stream_ptr()), "gelu_bwd")
    return dx


def attention(qkv, B, N, H, D=64):
    """qkv [B*N, 3*H*D] -> (out [B*N, H*D], lse [B,H,N])."""
    out = torch.empty((B * N, H * D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.load().ksmi_attention_forward(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, D, D ** -0.5,
                                                  DT[qkv.dtype], stream_ptr()), "attention_forward")
    return out, lse


def attention_backward(qkv, out, lse, dout, B, N, H, D=64):
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(_lib.load().ksmi_attention_bwd_workspace(B, N, H, D, DT[qkv.dtype]), dtype=torch.uint8, device=qkv.device)
    _lib.check(_lib.load().ksmi_attention_backward(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr(), dqkv.data_ptr(),
                                                   ws.data_ptr(), B, N, H, D, D ** -0.5, DT[qkv.dtype], stream_ptr()), "attention_backward")
    return dqkv


# ---------------------------------------------------------------------------------------------------
# ChangeFormer glue ops (cformer.hip)
# ---------------------------------------------------------------------------------------------------
def im2col(x, KH, KW, stride, pad, nchw_image=False, dtype=None):
    """x NHWC (or the NCHW fp32 image) -> col [B, Ho, Wo, Kpad], k = c*KH*KW + tap."""
    if nchw_image:
        B, Cin, H, W = x.shape
    else:
        B, H, W, Cin = x.shape
        dtype = x.dtype
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    kc = 32 if dtype == torch.bfloat16 else 16
    Kpad = -(-(Cin * KH * KW) // kc) * kc
    out = torch.empty((B, Ho, Wo, Kpad), dtype=dtype, device=x.device)
    _lib.check(_lib.load().ksmi_im2col(x.data_ptr(), out.data_ptr(), B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad,
                                       1 if nchw_image else 0, DT[dtype], stream_ptr()), "im2col")
    return out


def col2im(dcol, Cin, H, W, KH, KW, stride, pad, out=None):
    B, Ho, Wo, Kpad = dcol.shape
    acc = 0 if out is None else 1
    if out is None:
        out = torch.empty((B, H, W, Cin), dtype=dcol.dtype, device=dcol.device)
    _lib.check(_lib.load().ksmi_col2im(dcol.data_ptr(), out.data_ptr(), acc, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad,
                                       DT[dcol.dtype], stream_ptr()), "col2im")
    return out


def dwconv3x3_gelu(x, weight, bias):
    B, H, W, Cc = x.shape
    z, g = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_lib.load().ksmi_dwconv3x3_gelu_forward(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), z.data_ptr(), g.data_ptr(),
                                                       B, H, W, Cc, DT[x.dtype], stream_ptr()), "dwconv_fwd")
    return z, g


def dwconv3x3_backward(x, dz, weight):
    """(dx, dweight [C,1,3,3], dbias [C]) of z = depthwise3x3(x) + b given dz."""
    B, H, W, Cc = x.shape
    lib = _lib.load()
    dx = torch.empty_like(x)
    _lib.check(lib.ksmi_dwconv3x3_backward_input(dz.data_ptr(), weight.data_ptr(), dx.data_ptr(), B, H, W, Cc, DT[x.dtype], stream_ptr()), "dwconv_bwd")
    rows = max(1, min(1024, B * H * W // 16))
    partial = torch.empty((rows, 10 * Cc), dtype=torch.float32, device=x.device)
    _lib.check(lib.ksmi_dwconv3x3_wgrad(x.data_ptr(), dz.data_ptr(), partial.data_ptr(), rows, B, H, W, Cc, DT[x.dtype], stream_ptr()), "dwconv_wgrad")
    dw = torch.empty((Cc, 1, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty(Cc, dtype=torch.float32, device=x.device)
    _lib.check(lib.ksmi_reduce_rows(partial.data_ptr(), rows, 1, 10 * Cc, 9 * Cc, None, None, dw.data_ptr(), 0, stream_ptr()), "reduce_rows")
    _lib.check(lib.ksmi_reduce_rows(partial.data_ptr() + 9 * Cc * 4, rows, 1, 10 * Cc, Cc, None, None, db.data_ptr(), 0, stream_ptr()), "reduce_rows")
    return dx, dw, db


def _drop_args(p, site, rng_state):
    """(thr, inv_keep, site, state pointer) of one stochastic layer; p = 0 -> off (see ksmi.h: ksmi_dropout_apply)"""
    if p <= 0.0:
        return 0, 1.0, 0, None
    if p >= 1.0:
        raise ValueError("dropout probability must be < 1 (nn.Dropout(1.0) zeroes everything: drop the branch instead)")
    if rng_state is None or rng_state.dtype != torch.int32 or rng_state.numel() != 2:
        raise ValueError("rng_state: int32[2] device tensor {seed, step}")
    return min(0xFFFFFFFF, int(round(p * 4294967296.0))), 1.0 / (1.0 - p), site, rng_state.data_ptr()


def sr_attention(q, kv, B, Nq, Nk, heads, p=0.0, site=0, rng_state=None):
    Cc = q.shape[1]
    out = torch.empty_like(q)
    thr, inv, site, st = _drop_args(p, site, rng_state)
    _lib.check(_lib.load().ksmi_sr_attention_forward_drop(q.data_ptr(), kv.data_ptr(), out.data_ptr(), B, Nq, Nk, heads, Cc,
                                                          (Cc // heads) ** -0.5, thr, inv, site, st, DT[q.dtype], stream_ptr()), "sr_attention_fwd")
    return out


def sr_attention_backward(q, kv, out, dout, B, Nq, Nk, heads, p=0.0, site=0, rng_state=None):
    Cc = q.shape[1]
    lib = _lib.load()
    ws = torch.empty(lib.ksmi_sr_attention_bwd_workspace(B, Nq, Nk, heads, Cc), dtype=torch.uint8, device=q.device)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    thr, inv, site, st = _drop_args(p, site, rng_state)
    _lib.check(lib.ksmi_sr_attention_backward_drop(q.data_ptr(), kv.data_ptr(), out.data_ptr(), dout.data_ptr(), dq.data_ptr(), dkv.data_ptr(),
                                                   ws.data_ptr(), B, Nq, Nk, heads, Cc, (Cc // heads) ** -0.5, thr, inv, site, st, DT[q.dtype],
                                                   stream_ptr()), "sr_attention_bwd")
    return dq, dkv


def dropout_apply(x, rng_state, p=0.0, site=0, path_p=0.0, path_site=0, rows_per_sample=1, resid=None, out=None):
    """y = [resid +] x * Dropout(p; site) * DropPath(path_p; path_site) over a [rows, cols] matrix (ksmi_dropout_apply)"""
    rows, cols = x.shape
    y = torch.empty_like(x) if out is None else out
    t1, i1, s1, st1 = _drop_args(p, site, rng_state)
    t2, i2, s2, st2 = _drop_args(path_p, path_site, rng_state)
    _lib.check(_lib.load().ksmi_dropout_apply(x.data_ptr(), None if resid is None else resid.data_ptr(), y.data_ptr(), rows, cols, rows_per_sample,
                                              t1, i1, s1, t2, i2, s2, st1 or st2, DT[x.dtype], stream_ptr()), "dropout_apply")
    return y


def bilinear(x, Ho, Wo, add=None):
    B, Hi, Wi, Cc = x.shape
    y = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().ksmi_bilinear_forward(x.data_ptr(), None if add is None else add.data_ptr(), y.data_ptr(), B, Hi, Wi, Ho, Wo, Cc,
                                                 DT[x.dtype], stream_ptr()), "bilinear_fwd")
    return y


def bilinear_backward(dy, Hi, Wi, out=None):
    B, Ho, Wo, Cc = dy.shape
    acc = 0 if out is None else 1
    if out is None:
        out = torch.empty((B, Hi, Wi, Cc), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().ksmi_bilinear_backward(dy.data_ptr(), out.data_ptr(), acc, B, Hi, Wi, Ho, Wo, Cc, DT[dy.dtype], stream_ptr()), "bilinear_bwd")
    return out


# ---------------------------------------------------------------------------------------------------
# token GEMMs on the 128x128 MFMA tiles (gemm.hip, bf16)
# ---------------------------------------------------------------------------------------------------
def cast_bf16(w):
    out = torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.load().ksmi_cast_bf16(w.data_ptr(), out.data_ptr(), w.numel(), stream_ptr()), "cast_bf16")
    return out


def gemm_nt(x, wb, bias=None, resid=None):
    rows, K = x.shape
    N = wb.shape[0]
    y = torch.empty((rows, N), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().ksmi_gemm_nt(x.data_ptr(), K, wb.data_ptr(), K, None if bias is None else bias.data_ptr(),
                                        None if resid is None else resid.data_ptr(), N, y.data_ptr(), N, rows, K, N, stream_ptr()), "gemm_nt")
    return y


def gemm_nn(dy, wb, out=None):
    rows, N = dy.shape
    K = wb.shape[1]
    acc = 0 if out is None else 1
    if out is None:
        out = torch.empty((rows, K), dtype=torch.bfloat16, device=dy.device)
    _lib.check(_lib.load().ksmi_gemm_nn(dy.data_ptr(), N, wb.data_ptr(), K, out.data_ptr(), K, rows, K, N, acc, stream_ptr()), "gemm_nn")
    return out
