"""Functional (allocate-and-run) wrappers over the C-ABI conv family; used by the tests and
by one-off callers.  The model plans (snunet_plan.py) build the same descriptors once and
replay them instead."""
import ctypes as C

import torch

from . import _lib
from .runtime import DT, SrcSpec, conv_grid_m, make_conv, make_pack, make_wgrad, packed_weight_numel, require_gpu, stream_ptr


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
    Npad = (N + 15) // 16 * 16
    out = torch.empty(packed_weight_numel(table, taps, Npad, dtype), dtype=dtype, device=weight.device)
    d = make_pack(weight, out, table, taps, N, Npad, n_mod, sK, sN, sD, sT, flip)
    _lib.check(_lib.load().ksmi_pack_weights(C.byref(d), DT[dtype], stream_ptr()), "pack_weights")
    return out


def conv3x3(xs, weight, bias=None, affine=None, want_stats=False):
    """xs: list of NHWC tensors (virtual concat); weight [N, sum C, 3, 3] fp32; affine=(scale, shift, relu)
    applies to a single source.  Returns (out NHWC, stats[rows,2,Npad] or None)."""
    dtype = xs[0].dtype
    B, H, W, _ = xs[0].shape
    N = weight.shape[0]
    Ktot = sum(x.shape[3] for x in xs)
    srcs = [SrcSpec(x, x.shape[3]) for x in xs]
    if affine is not None:
        srcs[0].scale, srcs[0].shift, srcs[0].relu = affine
    out = torch.empty((B, H, W, N), dtype=dtype, device=xs[0].device)
    d, table = make_conv(srcs, [(out, N, 0, 0, N, 0)], out, bias, None, B, H, W, H, W, 3, 3, 1, 1, N, dtype)
    wpk = _pack(weight.contiguous(), table, 9, N, N, 9, Ktot * 9, 0, 1, 0, dtype)
    d.wpk = wpk.data_ptr()
    stats = None
    if want_stats:
        stats = torch.zeros((conv_grid_m(d), 2, d.Npad), dtype=torch.float32, device=out.device)
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


def conv3x3_wgrad(xs, dy, affine=None):
    dtype = dy.dtype
    B, H, W, N = dy.shape
    Ktot = sum(x.shape[3] for x in xs)
    srcs = [SrcSpec(x, x.shape[3]) for x in xs]
    if affine is not None:
        srcs[0].scale, srcs[0].shift, srcs[0].relu = affine
    grad = torch.zeros((N, Ktot, 3, 3), dtype=torch.float32, device=dy.device)
    d, ws = make_wgrad(srcs, dy, N, 0, N, grad, 9, Ktot * 9, 1, 0, B, H, W, H, W, 3, 3, 1, 1, dtype)
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
