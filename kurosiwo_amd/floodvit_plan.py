"""Static launch plan of one FloodViT forward/backward at a fixed (B, dtype).

Token activations are [rows = B*197][C] in the activation dtype (an NHWC image of rows x 1 pixels), so every
nn.Linear is a 1x1 implicit GEMM on ksmi_conv_forward / ksmi_conv_wgrad; the Decoder's ConvTranspose2d(k4,s2,p1)
layers run as four 2x2 phase convolutions (forward), one 4x4 stride-2 convolution (input gradient) and one 4x4
stride-2 weight-gradient GEMM.

Reference computation: /root/reference/models/vision_transformer.py:139-153 (ViT.forward), :35-66 (Attention),
:19-32 (FeedForward), :69-89 (Transformer); /root/reference/models/model_utilities.py:80-94
(FinetunerSegmentation.forward), :36-48 (Decoder.forward).
"""
import ctypes as C

import torch

from . import _lib
from .runtime import DT, SrcSpec, make_conv, make_pack, make_wgrad, packed_weight_numel
from .snunet_plan import LaunchList

LN_EPS = 1e-5


class FloodViTPlan:
    input_names = ("x",)

    def __init__(self, model, B, dtype, with_backward):
        self.m, self.B, self.dtype, self.with_backward = model, B, dtype, with_backward
        self.dev = model.flat_params.device
        self.dt = DT[dtype]
        self.lib = _lib.load()
        self.packs, self.fwd, self.bwd = LaunchList(), LaunchList(), LaunchList()
        self.keep, self._pack_descs, self._pinit = [], [], set()
        self.param_ready = {}
        self.named = {}            # debug/test access to intermediate activations
        self._need, self._bufs, self._later = {}, {}, []
        hp = model.hp
        self.D, self.depth, self.heads, self.mlp = hp["dim"], hp["depth"], hp["heads"], hp["mlp_dim"]
        self.inner = hp["heads"] * hp["dim_head"]
        self.Cin = hp["channels"]
        self.ih, self.iw = hp["image_size"]
        self.P = hp["patch_size"][0]
        self.gh, self.gw = model.grid
        self.npatch = self.gh * self.gw
        self.N1 = self.npatch + 1
        self.R = B * self.N1                 # token rows incl. cls
        self.Rp = B * self.npatch
        self.ncls = model.num_classes
        self.Cs = 8                          # channel stride of the NHWC logits (vector-aligned pad channels)
        self.train_encoder = with_backward and any(model._param_obj(k).requires_grad for k in model._pspec if k.startswith("model."))
        self.x = torch.empty((B, self.Cin, self.ih, self.iw), dtype=torch.float32, device=self.dev)
        self.logits = torch.empty((B, self.ncls, self.ih, self.iw), dtype=torch.float32, device=self.dev)
        self.dlogits = torch.empty_like(self.logits) if with_backward else None
        cmax = max(self.D, 128)
        self.const = torch.zeros((2, cmax), dtype=torch.float32, device=self.dev)   # row 0 = zeros, row 1 = ones
        self.const[1].fill_(1.0)
        self._build()
        if self._pack_descs:
            n = len(self._pack_descs)
            arr = (_lib.PackDesc * n)(*self._pack_descs)
            raw = bytes(C.string_at(C.addressof(arr), C.sizeof(arr)))
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
            self.keep.append(table)
            self.packs.add("ksmi_pack_weights_batched", lambda: (table.data_ptr(), n, self.dt),
                           {"kind": "pack_weights", "bytes": 0, "flops": 0})
        for name, nbytes in self._need.items():
            self._bufs[name] = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.dev)
        for fn in self._later:
            fn()
        for ll in (self.packs, self.fwd, self.bwd):
            ll.resolve(self.lib)

    # ---------------------------------------------------------------- small helpers
    def need(self, name, nbytes):
        self._need[name] = max(self._need.get(name, 0), int(nbytes))

    def scr(self, name):
        return self._bufs[name].data_ptr()

    def buf(self, *shape):
        t = torch.zeros(shape, dtype=self.dtype, device=self.dev)
        self.keep.append(t)
        return t

    def fbuf(self, *shape):
        t = torch.zeros(shape, dtype=torch.float32, device=self.dev)
        self.keep.append(t)
        return t

    def _es(self):
        return 2 if self.dtype == torch.bfloat16 else 4

    def _acc_param(self, key):
        acc = 1 if key in self._pinit else 0
        self._pinit.add(key)
        return acc

    def _mark(self, *keys):
        for k in keys:
            self.param_ready[k] = len(self.bwd.pending) - 1

    def _packed(self, key, table, taps, N, n_mod, sK, sN, sD, sT, flip=0, tap_map=None):
        Npad = (N + 15) // 16 * 16
        out = torch.empty(packed_weight_numel(table, taps, Npad, self.dtype), dtype=self.dtype, device=self.dev)
        d = make_pack(self.m._p(key), out, table, taps, N, Npad, n_mod, sK, sN, sD, sT, flip, tap_map)
        self.keep += [d, out]
        self._pack_descs.append(d)
        return out

    def _conv(self, ll, d, tag, name=""):
        self.keep.append(d)
        taps, es = d.KH * d.KW, self._es()
        ktot = sum(d.src[i].c_len for i in range(d.nsrc))
        pin, pout = d.B * d.Hin * d.Win, d.B * d.Hout * d.Wout
        elems = pin * ktot + sum(pout * d.dst[i].n_len * (2 if d.dst[i].accumulate else 1) for i in range(d.ndst))
        if d.mask_src:
            elems += pout * d.N
        meta = {"kind": f"igemm_{tag}<{d.KH}x{d.KW}s{d.stride}>", "bytes": elems * es + taps * ktot * d.N * es,
                "flops": 2 * pout * d.N * ktot * taps, "tag": f"{name} K={ktot} N={d.N} M={pout}"}
        ll.add("ksmi_conv_forward", lambda: (C.byref(d), self.dt), meta)

    def _wgrad(self, d, ws, key):
        self.keep.append(d)
        self.need("wgrad", ws)
        self._later.append(lambda: setattr(d, "partial", self.scr("wgrad")))
        taps, es = d.KH * d.KW, self._es()
        ktot = sum(d.src[i].c_len for i in range(d.nsrc))
        pin, pout = d.B * d.Hin * d.Win, d.B * d.Hout * d.Wout
        meta = {"kind": f"igemm_wgrad<{d.KH}x{d.KW}s{d.stride}>", "bytes": (pin * ktot + pout * d.N) * es + taps * ktot * d.N * 4,
                "flops": 2 * pout * d.N * ktot * taps, "tag": f"{key} K={ktot} N={d.N} M={pout}"}
        self.bwd.add("ksmi_conv_wgrad", lambda: (C.byref(d), self.dt), meta)
        self._mark(key)

    def _elt_meta(self, kind, nelem_rw):
        return {"kind": kind, "bytes": int(nelem_rw) * self._es(), "flops": 0}

    # ---------------------------------------------------------------- nn.Linear on token rows
    def _linear(self, name, x, Cin, wkey, bkey, out, N, rows):
        d, table = make_conv([SrcSpec(x, Cin)], [(out, N, 0, 0, N, 0)], out, self.m._p(bkey) if bkey else None, None,
                             1, rows, 1, rows, 1, 1, 1, 1, 0, N, self.dtype)
        d.wpk = self._packed(wkey, table, 1, N, N, 1, Cin, 0, 0).data_ptr()
        self._conv(self.fwd, d, "linear", name)

    def _bias_grad(self, dy, rows, N, bkey):
        r = max(1, min(512, rows // 64))
        self.need("red", r * N * 4)
        acc = self._acc_param(bkey)
        gb = self.m._g(bkey).data_ptr()
        self.bwd.add("ksmi_channel_sum", lambda: (dy.data_ptr(), self.scr("red"), r, rows, N, self.dt),
                     self._elt_meta("channel_sum", rows * N))
        self.bwd.add("ksmi_reduce_rows", lambda: (self.scr("red"), r, 1, N, N, None, None, gb, acc))
        self._mark(bkey)

    def _linear_bwd(self, name, x, Cin, wkey, bkey, dy, N, rows, dx, want_w=True):
        """dx = dy @ W ("="; skipped if dx is None) ; dW = dy^T x ; db = colsum(dy)"""
        if dx is not None:
            d, table = make_conv([SrcSpec(dy, N)], [(dx, Cin, 0, 0, Cin, 0)], dx, None, None,
                                 1, rows, 1, rows, 1, 1, 1, 1, 0, Cin, self.dtype)
            d.wpk = self._packed(wkey, table, 1, Cin, Cin, Cin, 1, 0, 0).data_ptr()
            self._conv(self.bwd, d, "linear_dgrad", name)
        if want_w:
            dw, ws = make_wgrad([SrcSpec(x, Cin)], dy, N, 0, N, self.m._g(wkey), 1, Cin, 0, self._acc_param(wkey),
                                1, rows, 1, rows, 1, 1, 1, 1, 0, self.dtype)
            self._wgrad(dw, ws, wkey)
            if bkey:
                self._bias_grad(dy, rows, N, bkey)

    # ---------------------------------------------------------------- nn.LayerNorm
    def _ln(self, x, wkey, bkey, y, rows, Cc):
        st = self.fbuf(2, rows)
        g, b = self.m._p(wkey).data_ptr(), self.m._p(bkey).data_ptr()
        self.fwd.add("ksmi_layernorm_forward", lambda: (x.data_ptr(), g, b, y.data_ptr(), st[0].data_ptr(), st[1].data_ptr(),
                                                        rows, Cc, LN_EPS, self.dt), self._elt_meta("layernorm_fwd", 2 * rows * Cc))
        return st

    def _ln_bwd(self, dy, x, st, wkey, bkey, dx, accumulate, rows, Cc, want_w=True):
        nblk = self.lib.ksmi_layernorm_bwd_blocks(rows)
        self.need("lnp", nblk * 2 * Cc * 4)
        g = self.m._p(wkey).data_ptr()
        self.bwd.add("ksmi_layernorm_backward", lambda: (dy.data_ptr(), x.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), g,
                                                         dx.data_ptr(), accumulate, self.scr("lnp"), rows, Cc, self.dt),
                     self._elt_meta("layernorm_bwd", (3 + accumulate) * rows * Cc))
        if want_w:
            a1, a2 = self._acc_param(wkey), self._acc_param(bkey)
            assert a1 == a2
            gw, gb = self.m._g(wkey).data_ptr(), self.m._g(bkey).data_ptr()
            self.bwd.add("ksmi_reduce_rows", lambda: (self.scr("lnp"), nblk, 2, Cc, Cc, None, gw, gb, a1))
            self._mark(wkey, bkey)

    # ---------------------------------------------------------------- ConvTranspose2d(k4, s2, p1)
    def _deconv(self, name, x, Cin, N, H, W, out, outC, k_real_out=None):
        """out[B,2H,2W,outC][..., :N] = ConvTranspose2d(x) + bias as 4 phase convolutions with 2x2 taps:
        out[2m+py] = sum_a x[m - pad + a] * W[ky],  pad = 1 - py,  ky = (3 - 2a) if py == 0 else (2 - 2a)."""
        wkey, bkey = f"head.{name}.weight", f"head.{name}.bias"
        B = self.B
        for py in range(2):
            for px in range(2):
                tap_map = []
                for a in range(2):
                    for b in range(2):
                        ky = 3 - 2 * a if py == 0 else 2 - 2 * a
                        kx = 3 - 2 * b if px == 0 else 2 - 2 * b
                        tap_map.append(ky * 4 + kx)
                d, table = make_conv([SrcSpec(x, Cin)], [(out, outC, 0, 0, N, 0)], out, self.m._p(bkey), None,
                                     B, H, W, H, W, 2, 2, 1, 1 - py, N, self.dtype, pad_x=1 - px,
                                     out_map=(2, 2, py, px, 2 * H, 2 * W))
                # Wt[c][n][ky][kx]: k = c, column = n
                d.wpk = self._packed(wkey, table, 4, N, N, N * 16, 16, 0, 1, 0, tap_map).data_ptr()
                self._conv(self.fwd, d, "deconv_phase", f"{name}.p{py}{px}")

    def _deconv_bwd(self, name, x, Cin, N, H, W, dout, doutC, dx, mask=None):
        """dout [B,2H,2W,doutC] (first N channels real).  dx[B,H,W,Cin] = 4x4 stride-2 conv of dout (optionally
        ReLU-masked by `mask`), dW via the stride-2 weight-gradient GEMM, db = channel sums."""
        wkey, bkey = f"head.{name}.weight", f"head.{name}.bias"
        B = self.B
        src = [SrcSpec(dout, doutC, 0, doutC, k_real=N)]
        if dx is not None:
            mk = None
            if mask is not None:
                mk = (mask, self.const[0], self.const[1], self.const[1], self.const[0])
            d, table = make_conv(src, [(dx, Cin, 0, 0, Cin, 0)], dx, None, None, B, 2 * H, 2 * W, H, W, 4, 4, 2, 1, Cin,
                                 self.dtype, mask=mk)
            d.wpk = self._packed(wkey, table, 16, Cin, Cin, 16, N * 16, 0, 1, 0).data_ptr()
            self._conv(self.bwd, d, "deconv_dgrad", name)
        dw, ws = make_wgrad(src, x, Cin, 0, Cin, self.m._g(wkey), 16, N * 16, 1, self._acc_param(wkey),
                            B, 2 * H, 2 * W, H, W, 4, 4, 2, 1, self.dtype)
        self._wgrad(dw, ws, wkey)
        # bias gradient over the real channels only
        rows = B * 4 * H * W
        r = max(1, min(512, rows // 256))
        self.need("red", r * doutC * 4)
        acc = self._acc_param(bkey)
        gb = self.m._g(bkey).data_ptr()
        self.bwd.add("ksmi_channel_sum", lambda: (dout.data_ptr(), self.scr("red"), r, rows, doutC, self.dt),
                     self._elt_meta("channel_sum", rows * doutC))
        self.bwd.add("ksmi_reduce_rows", lambda: (self.scr("red"), r, 1, doutC, N, None, None, gb, acc))
        self._mark(bkey)

    # ---------------------------------------------------------------- the graph
    def _build(self):
        m, B, D, R, Rp, I, M = self.m, self.B, self.D, self.R, self.Rp, self.inner, self.mlp
        dt, dtype = self.dt, self.dtype
        pd = self.Cin * self.P * self.P
        bwd_steps = []        # closures appended in forward order, executed reversed

        # ---- patch embedding (vision_transformer.py:121-126, 140) -------------------------------
        P0, P1, E0, E1 = self.buf(Rp, pd), self.buf(Rp, pd), self.buf(Rp, D), self.buf(Rp, D)
        self.fwd.add("ksmi_patchify", lambda: (self.x.data_ptr(), P0.data_ptr(), B, self.Cin, self.ih, self.iw, self.P, dt),
                     self._elt_meta("patchify", 3 * Rp * pd))
        k = "model.to_patch_embedding"
        st_p1 = self._ln(P0, f"{k}.1.weight", f"{k}.1.bias", P1, Rp, pd)
        self._linear("patch_embed", P1, pd, f"{k}.2.weight", f"{k}.2.bias", E0, D, Rp)
        st_p3 = self._ln(E0, f"{k}.3.weight", f"{k}.3.bias", E1, Rp, D)
        X = self.buf(R, D)
        self.named.update(patches=P0, embed=E1, x0=X)
        cls, pos = m._p("model.cls_token").data_ptr(), m._p("model.pos_embedding").data_ptr()
        self.fwd.add("ksmi_vit_embed_forward", lambda X0=X: (E1.data_ptr(), cls, pos, X0.data_ptr(), B, self.N1, D, dt),
                     self._elt_meta("vit_embed", 2 * R * D))

        if self.with_backward:
            gx = self.buf(R, D)              # gradient of the residual stream, updated in place layer by layer
            tD, tI, tM, tQ = self.buf(R, D), self.buf(R, I), self.buf(R, M), self.buf(R, 3 * I)
        else:
            gx = tD = tI = tM = tQ = None
        t1 = self.buf(R, D)

        def embed_bwd():
            dE1, dE0, dP1 = self.buf(Rp, D), self.buf(Rp, D), self.buf(Rp, pd)
            gcls, gpos = m._g("model.cls_token").data_ptr(), m._g("model.pos_embedding").data_ptr()
            self.bwd.add("ksmi_vit_embed_backward", lambda: (gx.data_ptr(), dE1.data_ptr(), gcls, gpos, B, self.N1, D, dt),
                         self._elt_meta("vit_embed_bwd", 2 * R * D))
            self._pinit.update(["model.cls_token", "model.pos_embedding"])
            self._mark("model.cls_token", "model.pos_embedding")
            self._ln_bwd(dE1, E0, st_p3, f"{k}.3.weight", f"{k}.3.bias", dE0, 0, Rp, D)
            self._linear_bwd("patch_embed", P1, pd, f"{k}.2.weight", f"{k}.2.bias", dE0, D, Rp, dP1)
            self._ln_bwd(dP1, P0, st_p1, f"{k}.1.weight", f"{k}.1.bias", P1, 0, Rp, pd)   # dx of the raw patches is unused
        bwd_steps.append(embed_bwd)

        # ---- transformer (vision_transformer.py:84-89) ------------------------------------------
        for li in range(self.depth):
            a, f = f"model.transformer.layers.{li}.0", f"model.transformer.layers.{li}.1"
            x_in = X
            h1, qkv, att, x_mid = self.buf(R, D), self.buf(R, 3 * I), self.buf(R, I), self.buf(R, D)
            h2, u, g, x_out = self.buf(R, D), self.buf(R, M), self.buf(R, M), self.buf(R, D)
            lse = self.fbuf(B, self.heads, self.N1)
            scale = float(self.m.hp["dim_head"]) ** -0.5
            st1 = self._ln(x_in, f"{a}.norm.weight", f"{a}.norm.bias", h1, R, D)
            self._linear(f"L{li}.to_qkv", h1, D, f"{a}.to_qkv.weight", None, qkv, 3 * I, R)
            aflops = 4 * B * self.heads * self.N1 * self.N1 * 64
            self.fwd.add("ksmi_attention_forward", lambda qkv=qkv, att=att, lse=lse: (
                qkv.data_ptr(), att.data_ptr(), lse.data_ptr(), B, self.N1, self.heads, 64, scale, dt),
                {"kind": "attention_fwd", "bytes": 4 * R * I * self._es(), "flops": aflops})
            self._linear(f"L{li}.to_out", att, I, f"{a}.to_out.0.weight", f"{a}.to_out.0.bias", t1, D, R)
            self.fwd.add("ksmi_add", lambda x_in=x_in, x_mid=x_mid: (t1.data_ptr(), x_in.data_ptr(), x_mid.data_ptr(), R * D, dt),
                         self._elt_meta("add", 3 * R * D))
            st2 = self._ln(x_mid, f"{f}.net.0.weight", f"{f}.net.0.bias", h2, R, D)
            self._linear(f"L{li}.ff1", h2, D, f"{f}.net.1.weight", f"{f}.net.1.bias", u, M, R)
            self.fwd.add("ksmi_gelu_forward", lambda u=u, g=g: (u.data_ptr(), g.data_ptr(), R * M, dt), self._elt_meta("gelu", 2 * R * M))
            self._linear(f"L{li}.ff2", g, M, f"{f}.net.4.weight", f"{f}.net.4.bias", t1, D, R)
            self.fwd.add("ksmi_add", lambda x_mid=x_mid, x_out=x_out: (t1.data_ptr(), x_mid.data_ptr(), x_out.data_ptr(), R * D, dt),
                         self._elt_meta("add", 3 * R * D))
            X = x_out
            self.named[f"layer{li}"] = x_out

            def layer_bwd(li=li, a=a, f=f, x_in=x_in, h1=h1, qkv=qkv, att=att, x_mid=x_mid, h2=h2, u=u, g=g, lse=lse,
                          st1=st1, st2=st2, scale=scale, aflops=aflops):
                # FeedForward: x_out = x_mid + W2 gelu(W1 LN(x_mid) + b1) + b2
                self._linear_bwd(f"L{li}.ff2", g, M, f"{f}.net.4.weight", f"{f}.net.4.bias", gx, D, R, tM)
                self.bwd.add("ksmi_gelu_backward", lambda: (tM.data_ptr(), u.data_ptr(), tM.data_ptr(), R * M, dt),
                             self._elt_meta("gelu_bwd", 3 * R * M))
                self._linear_bwd(f"L{li}.ff1", h2, D, f"{f}.net.1.weight", f"{f}.net.1.bias", tM, M, R, tD)
                self._ln_bwd(tD, x_mid, st2, f"{f}.net.0.weight", f"{f}.net.0.bias", gx, 1, R, D)
                # Attention: x_mid = x_in + Wo attn(Wqkv LN(x_in)) + bo
                self._linear_bwd(f"L{li}.to_out", att, I, f"{a}.to_out.0.weight", f"{a}.to_out.0.bias", gx, D, R, tI)
                self.bwd.add("ksmi_attention_backward", lambda: (qkv.data_ptr(), att.data_ptr(), lse.data_ptr(), tI.data_ptr(),
                                                                 tQ.data_ptr(), B, self.N1, self.heads, 64, scale, dt),
                             {"kind": "attention_bwd", "bytes": 8 * R * I * self._es(), "flops": 5 * aflops // 2})
                self._linear_bwd(f"L{li}.to_qkv", h1, D, f"{a}.to_qkv.weight", None, tQ, 3 * I, R, tD)
                self._ln_bwd(tD, x_in, st1, f"{a}.norm.weight", f"{a}.norm.bias", gx, 1, R, D)
            bwd_steps.append(layer_bwd)

        # ---- final LN, drop cls, Decoder (vision_transformer.py:89,150-151; model_utilities.py:85-93,36-48) ----
        XF, F = self.buf(R, D), self.buf(Rp, D)
        x_last = X
        st_f = self._ln(x_last, "model.transformer.norm.weight", "model.transformer.norm.bias", XF, R, D)
        self.fwd.add("ksmi_drop_cls", lambda: (XF.data_ptr(), F.data_ptr(), B, self.N1, D, 0, dt), self._elt_meta("drop_cls", 2 * Rp * D))
        gh, gw = self.gh, self.gw
        D1 = self.buf(B, 2 * gh, 2 * gw, 128)
        U1 = self.buf(B, 4 * gh, 4 * gw, 128)
        D2 = self.buf(B, 8 * gh, 8 * gw, 64)
        L = self.buf(B, 16 * gh, 16 * gw, self.Cs)
        self.named.update(xf=XF, feat=F, d1=D1, u1=U1, d2=D2, logits_nhwc=L)
        self._deconv("deconv1", F, D, 128, gh, gw, D1, 128)
        self.fwd.add("ksmi_upsample2_forward", lambda: (D1.data_ptr(), U1.data_ptr(), B, 2 * gh, 2 * gw, 128, 1, dt),
                     self._elt_meta("upsample2", 5 * D1.numel()))
        self._deconv("deconv2", U1, 128, 64, 4 * gh, 4 * gw, D2, 64)
        self.fwd.add("ksmi_relu_forward", lambda: (D2.data_ptr(), D2.data_ptr(), D2.numel(), dt), self._elt_meta("relu", 2 * D2.numel()))
        self._deconv("deconv3", D2, 64, self.ncls, 8 * gh, 8 * gw, L, self.Cs)
        HW = self.ih * self.iw
        if (16 * gh, 16 * gw) != (self.ih, self.iw):
            raise _lib.KsmiError("FloodViT decoder expects patch size 16 (14 -> 28 -> 56 -> 112 -> 224, model_utilities.py:36-48)")
        self.fwd.add("ksmi_logits_to_nchw", lambda: (L.data_ptr(), self.logits.data_ptr(), B, self.ncls, self.Cs, HW, dt),
                     self._elt_meta("logits_to_nchw", B * HW * (self.ncls + 2 * self.ncls)))

        if not self.with_backward:
            return
        # ---- backward ---------------------------------------------------------------------------
        dL = self.buf(B, self.ih, self.iw, self.Cs)
        dD2, dU1, dD1, dF = self.buf(*D2.shape), self.buf(*U1.shape), self.buf(*D1.shape), self.buf(Rp, D)
        self.bwd.add("ksmi_dlogits_to_nhwc", lambda: (self.dlogits.data_ptr(), dL.data_ptr(), B, self.ncls, self.Cs, HW, dt),
                     self._elt_meta("dlogits_to_nhwc", B * HW * (2 * self.ncls + self.Cs)))
        self._deconv_bwd("deconv3", D2, 64, self.ncls, 8 * gh, 8 * gw, dL, self.Cs, dD2, mask=D2)
        self._deconv_bwd("deconv2", U1, 128, 64, 4 * gh, 4 * gw, dD2, 64, dU1)
        self.bwd.add("ksmi_upsample2_backward", lambda: (dU1.data_ptr(), D1.data_ptr(), dD1.data_ptr(), B, 2 * gh, 2 * gw, 128, 1, dt),
                     self._elt_meta("upsample2_bwd", 6 * D1.numel()))
        self._deconv_bwd("deconv1", F, D, 128, gh, gw, dD1, 128, dF if self.train_encoder else None)
        if not self.train_encoder:
            return
        self.bwd.add("ksmi_drop_cls", lambda: (dF.data_ptr(), tD.data_ptr(), B, self.N1, D, 1, dt), self._elt_meta("drop_cls_bwd", 2 * R * D))
        self._ln_bwd(tD, x_last, st_f, "model.transformer.norm.weight", "model.transformer.norm.bias", gx, 0, R, D)
        for step in reversed(bwd_steps):
            step()

    # ---------------------------------------------------------------- execution
    def run_forward(self, x):
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.packs.run()
        self.fwd.run()
        return self.logits

    def run_backward(self, dlogits=None):
        if not self.with_backward:
            raise _lib.KsmiError("plan was built without backward")
        if dlogits is not None and dlogits.data_ptr() != self.dlogits.data_ptr():
            self.dlogits.copy_(dlogits)
        self.bwd.run()
