"""Host-side plumbing shared by the plans: device buffers (PyTorch allocator), stream
handles, descriptor builders for the implicit-GEMM family, and launch lists.

PyTorch is used here only for device memory and streams; all math is in libksmi.so.
"""
import ctypes as C
import functools

import torch

from . import _lib
from ._lib import KSMI_BF16, KSMI_F32, ConvDesc, Dst, PackDesc, Src, WgradDesc, check

DT = {torch.float32: KSMI_F32, torch.bfloat16: KSMI_BF16}


def require_gpu(t):
    if not t.is_cuda:
        raise _lib.KsmiError("kurosiwo_amd runs on an MI355X (HIP) device only: got a CPU tensor. "
                             "There is no CPU fallback; use the reference repo for CPU runs.")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def chunk_elems(dtype):
    return 32 if dtype == torch.bfloat16 else 16


def vec_elems(dtype):
    return 8 if dtype == torch.bfloat16 else 4


class Act:
    """An NHWC activation (+ its gradient buffer) inside a plan."""

    def __init__(self, name, B, H, W, Cch, dtype, device, needs_grad=True):
        self.name, self.B, self.H, self.W, self.C = name, B, H, W, Cch
        self.t = torch.empty((B, H, W, Cch), dtype=dtype, device=device)
        self.g = None
        self.needs_grad = needs_grad
        self.ginit = False        # has some backward op already written self.g (plan-build-time tracking)
        self.consumers = []       # (di tensor, C_di, conv weight key, channel offset in that conv's K, Ktot) of 3x3 convs reading this
        self.dtype, self.device = dtype, device

    def grad(self):
        if self.g is None:
            self.g = torch.empty_like(self.t)
        return self.g

    def take_acc_flag(self):
        """accumulate flag for the next backward writer (first writer overwrites)."""
        acc = 1 if self.ginit else 0
        self.ginit = True
        return acc


class SrcSpec:
    def __init__(self, tensor, Cch, c_off=0, c_len=None, scale=None, shift=None, relu=0, k_real=None):
        self.tensor, self.C, self.c_off = tensor, Cch, c_off
        self.c_len = Cch - c_off if c_len is None else c_len
        # channels that exist in the weight (the rest of c_len are zero pad channels of the activation)
        self.k_real = self.c_len if k_real is None else k_real
        self.scale, self.shift, self.relu = scale, shift, relu


@functools.lru_cache(maxsize=None)
def choose_patch(H, W, stride=1, KH=3, KW=3, max_pix=256, max_halo=512):
    """Pick the output patch (TH, TW) of a workgroup: maximise useful pixels per 256-row
    M-tile, then minimise halo area."""
    import os
    if os.environ.get("KSMI_PATCH"):                      # experiment override "TH,TW"
        th, tw = (int(v) for v in os.environ["KSMI_PATCH"].split(","))
        return min(th, H), min(tw, W)
    best = None
    for tw in range(1, min(W, 256) + 1):
        for th in range(1, min(H, max_pix // tw) + 1):
            hp = ((th - 1) * stride + KH) * ((tw - 1) * stride + KW)
            if hp > max_halo:
                continue
            tiles = -(-H // th) * -(-W // tw)
            # a workgroup always issues the MFMAs of a full 256-row M tile (4 waves x 4 fragments x 16)
            cost = tiles * 256 * (1.0 + 0.15 * hp / 256.0)
            key = (cost, hp)
            if best is None or key < best[0]:
                best = (key, th, tw)
    return best[1], best[2]


def choose_patch_wgrad(H, W, stride=1, KH=3, KW=3, max_pix=256, max_halo=512):
    """Patch of the weight-gradient kernel.  Its bf16 fast path (k-step-invariant fragment addresses, csrc/igemm.hip LIN) needs a patch
    width that divides 16 and a whole number of 32-pixel k-steps; prefer such a patch when it does not need more patches than the
    default choice (28x28 maps: 28x8 instead of 14x14; 14x14 maps: 14x16, the two extra columns are masked)."""
    import os
    th0, tw0 = choose_patch(H, W, stride, KH, KW, max_pix, max_halo)
    if os.environ.get("KSMI_PATCH") or ((16 % tw0) == 0 and (th0 * tw0) % 32 == 0):
        return th0, tw0
    tiles0 = -(-H // th0) * -(-W // tw0)
    best = None
    for tw in (16, 8, 4):
        if tw >= 2 * W:
            continue
        for th in range(1, min(H, max_pix // tw) + 1):
            if (th * tw) % 32:
                continue
            hp = ((th - 1) * stride + KH) * ((tw - 1) * stride + KW)
            if hp > max_halo:
                continue
            tiles = -(-H // th) * -(-W // tw)
            key = (tiles, th * tw, hp)
            if tiles <= tiles0 and (best is None or key < best[0]):
                best = (key, th, tw)
    return (best[1], best[2]) if best else (th0, tw0)


def _chunk_table(srcs, kc):
    """[(src_idx, c0, k_global, k_len)] walking the virtual concat in kc-element chunks."""
    table, kbase = [], 0
    for si, s in enumerate(srcs):
        for c0 in range(0, s.c_len, kc):
            table.append((si, c0, kbase + c0, max(0, min(kc, s.k_real - c0))))
        kbase += s.k_real
    if len(table) > _lib.MAX_CHUNKS and len(srcs) != 1:
        raise _lib.KsmiError(f"too many k-chunks ({len(table)} > {_lib.MAX_CHUNKS}) for a multi-source descriptor")
    return table, kbase


def _uniform(table, kc):
    """Descriptors with more chunks than the tables hold use the computed (uniform) chunk walk of one source."""
    if len(table) <= _lib.MAX_CHUNKS:
        return 0, 0
    assert all(t[0] == 0 and t[1] == i * kc and t[2] == i * kc for i, t in enumerate(table))
    return kc, table[-1][2] + table[-1][3]


def _fill_srcs(desc, srcs):
    desc.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        d = desc.src[i]
        d.ptr = s.tensor.data_ptr()
        d.scale = s.scale.data_ptr() if s.scale is not None else None
        d.shift = s.shift.data_ptr() if s.shift is not None else None
        d.C, d.c_off, d.c_len, d.relu = s.C, s.c_off, s.c_len, s.relu


class Launches:
    """A flat list of prepared C-ABI calls (fn, args) executed in order on the current stream."""

    def __init__(self):
        self.calls = []
        self.keep = []       # keep descriptors / tensors alive

    def add(self, name, *args):
        fn = getattr(_lib.load(), name)
        self.calls.append((fn, args, name))

    def run(self):
        st = stream_ptr()
        for fn, args, name in self.calls:
            rc = fn(*args, st)
            if rc != 0:
                check(rc, name)


def make_pack(w, out, table, taps, N, Npad, n_mod, sK, sN, sD, sT, flip, tap_map=None):
    d = PackDesc()
    d.w, d.out = w.data_ptr(), out.data_ptr()
    d.nchunks, d.taps, d.N, d.Npad, d.n_mod = len(table), taps, N, Npad, n_mod
    d.sK, d.sN, d.sD, d.sT, d.flip = sK, sN, sD, sT, flip
    kc_u, ktot = _uniform(table, table[0][3] if len(table) == 1 else table[1][2] - table[0][2])
    if kc_u:
        d.uniform_kc, d.k_total = kc_u, ktot
    else:
        for i, (_, _, kg, kl) in enumerate(table):
            d.k_off[i], d.k_len[i] = kg, kl
    if tap_map is not None:
        d.use_tap_map = 1
        for i, t in enumerate(tap_map):
            d.tap_map[i] = t
    return d


def conv_npad(N):
    """Padded column count of a convolution's packed weights / statistics rows: a multiple of 16, and 32 for the thin heads (N < 16) so
    that the 32-column tiles of the persistent kernels can take them (igemm4.hip <8, 2>: ChangeFormer's 256 -> 2 change_probability)."""
    return 32 if N < 16 else (N + 15) // 16 * 16


def make_conv(srcs, dsts, wpk, bias, stats, B, Hin, Win, Hout, Wout, KH, KW, stride, pad, N, dtype,
              mask=None, ps_cout=0, max_pix=256, pad_x=None, out_map=None, alpha=0.0, relu_out=0, resid=None, in_map=None):
    """dsts: list of (tensor, C, c_off, n_begin, n_len, accumulate).  mask: (tensor, mean, rstd, scale, shift).
    resid: (tensor, C) added after alpha scaling; relu_out: ReLU before statistics/store."""
    kc = chunk_elems(dtype)
    table, _ = _chunk_table(srcs, kc)
    d = ConvDesc()
    _fill_srcs(d, srcs)
    d.ndst = len(dsts)
    for i, (t, Cc, c_off, nb, nl, acc) in enumerate(dsts):
        q = d.dst[i]
        q.ptr, q.C, q.c_off, q.n_begin, q.n_len, q.accumulate = t.data_ptr(), Cc, c_off, nb, nl, acc
    d.wpk = wpk.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.stats = stats.data_ptr() if stats is not None else None
    if mask is not None:
        d.mask_src, d.m_mean, d.m_rstd, d.m_scale, d.m_shift = [x.data_ptr() for x in mask]
    d.B, d.Hin, d.Win, d.Hout, d.Wout = B, Hin, Win, Hout, Wout
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    d.pad_x = pad if pad_x is None else pad_x
    if in_map is not None:           # (sy, sx, oy, ox, H, W): strided view of the source tensor
        d.in_sy, d.in_sx, d.in_oy, d.in_ox, d.in_H, d.in_W = in_map
    d.alpha, d.relu_out = alpha, relu_out
    if resid is not None:
        d.resid, d.residC = resid[0].data_ptr(), resid[1]
    if out_map is not None:          # (sy, sx, oy, ox, H, W)
        d.out_sy, d.out_sx, d.out_oy, d.out_ox, d.out_H, d.out_W = out_map
    mp = max_pix if stride == 1 else min(max_pix, 128)
    d.TH, d.TW = choose_patch(Hout, Wout, stride, KH, KW, mp)
    d.N, d.Npad = N, conv_npad(N)
    d.nchunks = len(table)
    d.ps_cout = ps_cout
    if _uniform(table, kc)[0]:
        d.uniform_kc = kc
    else:
        for i, (si, c0, _, _) in enumerate(table):
            d.chunk_src[i], d.chunk_c0[i] = si, c0
    return d, table


def conv_grid_m(d):
    return _lib.load().ksmi_conv_grid_m(C.byref(d))


def conv_stats_rows(d, dtype):
    """rows of the `stats` buffer of this descriptor (recorded in d.stats_rows: the persistent short-K kernel writes one row per
    workgroup, every other kernel one per M-tile); call after the descriptor is complete, with d.stats already non-null."""
    import os
    if os.environ.get("KSMI_STATS_ROWS_TILE") == "1":       # A/B switch: the tile kernel's row count (keeps a convolution with statistics on igemm2)
        return conv_grid_m(d)
    rows = _lib.load().ksmi_conv_stats_rows(C.byref(d), DT[dtype])
    d.stats_rows = rows
    return rows


def packed_weight_numel(table, taps, Npad, dtype):
    return len(table) * taps * Npad * chunk_elems(dtype)


def make_wgrad(srcs, dy, dyC, dy_c_off, N, grad, gK, gN, gT, accumulate, B, Hin, Win, Hout, Wout,
               KH, KW, stride, pad, dtype, pad_x=None, in_map=None, tap_off=None):
    kc = chunk_elems(dtype)
    table, _ = _chunk_table(srcs, kc)
    d = WgradDesc()
    _fill_srcs(d, srcs)
    d.dy, d.dyC, d.dy_c_off = dy.data_ptr(), dyC, dy_c_off
    d.B, d.Hin, d.Win, d.Hout, d.Wout = B, Hin, Win, Hout, Wout
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    mp = 256 if stride == 1 else 128
    d.TH, d.TW = choose_patch_wgrad(Hout, Wout, stride, KH, KW, mp)
    d.N, d.nchunks = N, len(table)
    if pad_x is not None:
        d.pad_x_set, d.pad_x = 1, pad_x
    if in_map is not None:
        d.in_sy, d.in_sx, d.in_oy, d.in_ox, d.in_H, d.in_W = in_map
    if tap_off is not None:
        d.use_tap_off = 1
        for i, t in enumerate(tap_off):
            d.tap_off[i] = t
    d.grad, d.gK, d.gN, d.gT, d.accumulate = grad.data_ptr(), gK, gN, gT, accumulate
    kc_u, ktot = _uniform(table, kc)
    if kc_u:
        d.uniform_kc, d.k_total = kc_u, ktot
    else:
        for i, (si, c0, kg, kl) in enumerate(table):
            d.chunk_src[i], d.chunk_c0[i], d.k_off[i], d.k_len[i] = si, c0, kg, kl
    lib = _lib.load()
    ws = lib.ksmi_conv_wgrad_workspace(C.byref(d), DT[dtype])
    npad = (N + 15) // 16 * 16
    slab = (KH * KW) * len(table) * kc * npad * 4
    d.nsplit = ws // slab
    return d, ws
