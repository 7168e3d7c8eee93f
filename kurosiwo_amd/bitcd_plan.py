"""Static launch plan of one BIT-CD (`base_resnet18`) forward/backward (row N2 of SURVEY.md §8).

Reference computation: /root/reference/models/bit_cd.py `ResNet.forward` :763-778 / `forward_single` :780-797.  The siamese ResNet-18
runs once per date through the BasicBlock builder of the U1 row (kurosiwo_amd/unet_plan.py: conv -> BN -> ReLU chains fused into the
consumers' operand loads and input-gradient epilogues); weight gradients of the second pass accumulate, BatchNorm statistics are
per date.  Head: nearest x2, conv_pred, |f1 - f2|, bilinear x4, conv3x3 -> BN -> ReLU -> conv3x3.
"""
import ctypes as C

import os

import torch

from .bitcd import LAYERS
from .changeformer_plan import CS
from .runtime import SrcSpec, conv_grid_m, conv_stats_rows, make_conv, make_wgrad
from .snunet_plan import _Saved
from .unet_plan import UnetPlan


class BitCDPlan(UnetPlan):
    input_names = ("x1", "x2")

    def __init__(self, model, B, H, W, dtype, training, with_backward):
        self._init_base(model, dtype, with_backward)
        self.B, self.H, self.W, self.training = B, H, W, training
        self.cin, self.nc = model.input_nc, model.output_nc
        self.x = torch.empty((2, B, self.cin, H, W), dtype=torch.float32, device=self.dev)
        self.xA, self.xB = self.x[0], self.x[1]
        self.logits = torch.empty((B, self.nc, H, W), dtype=torch.float32, device=self.dev)
        self.dlogits = torch.empty_like(self.logits) if with_backward else None
        self.const = torch.zeros((2, 512), dtype=torch.float32, device=self.dev)
        self.const[1].fill_(1.0)
        self._gbuf, self._gacc, self._bwd = {}, set(), []
        self._build_bitcd()
        if with_backward:
            for f in reversed(self._bwd):
                f()
        self._finish()

    # ---------------------------------------------------------------- conv1 7x7 s2 -> bn1 -> relu -> maxpool 3x3 s2 (one date)
    def _stem(self, date):
        B, H, W, dt = self.B, self.H, self.W, self.dt
        H1, W1 = H // 2, W // 2
        R1 = B * H1 * W1
        kc = 32 if self.dtype == torch.bfloat16 else 16
        Kreal = self.cin * 49
        Kpad = -(-Kreal // kc) * kc
        col, s0, f1 = self.buf(R1, Kpad), self.buf(R1, 64), self.buf(B, H1, W1, 64)
        sv0 = _Saved(64, self.dev)
        self.fwd.add("ksmi_im2col", lambda: (self.x[date].data_ptr(), col.data_ptr(), B, self.cin, H, W, H1, W1, 7, 7, 2, 3, Kpad, 1, dt),
                     self._elt_meta("im2col", 2 * R1 * Kpad))
        d, table = make_conv([SrcSpec(col, Kpad, k_real=Kreal)], [(s0, 64, 0, 0, 64, 0)], s0, None, None, 1, R1, 1, R1, 1, 1, 1, 1, 0, 64, self.dtype)
        d.wpk = self._packed("resnet.conv1.weight", table, 1, 64, 64, 1, Kreal, 0, 0).data_ptr()
        rows0 = conv_stats_rows(d, self.dtype) if self.training else conv_grid_m(d)   # (rows of the kernel that will run it: see changeformer_plan._conv3)
        if self.training:
            self.need("stats", rows0 * 2 * d.Npad * 4)
            self._later.append(lambda: setattr(d, "stats", self.scr("stats")))
        self._conv(self.fwd, d, "stem7x7", "resnet.conv1")
        self._bn_finalize("resnet.bn1", sv0, rows0, d.Npad, 64, R1)
        self._affine(self.fwd, s0, sv0, f1, R1, 64, 1)
        H2, W2 = H1 // 2, W1 // 2
        p = self.buf(B, H2, W2, 64)
        # with a backward pass the forward records the window position of each first maximum (one byte per output element): the backward
        # compares codes instead of re-reading up to four windows per input element (325 -> ~40 us at 112 x 112 x 64 x 32 images)
        pidx = torch.empty(p.numel(), dtype=torch.uint8, device=self.dev) if self.with_backward and not os.environ.get("KSMI_MAXPOOL_GATHER") else None
        if pidx is not None:
            self.fwd.add("ksmi_maxpool3x3s2_forward_idx", lambda: (f1.data_ptr(), p.data_ptr(), pidx.data_ptr(), B, H1, W1, 64, dt),
                         self._elt_meta("maxpool3", 2 * R1 * 64))
        else:
            self.fwd.add("ksmi_maxpool3x3s2_forward", lambda: (f1.data_ptr(), p.data_ptr(), B, H1, W1, 64, dt), self._elt_meta("maxpool3", 2 * R1 * 64))

        def bwd():
            df1, ds0 = self.gbuf(f1), self.buf(R1, 64)
            dp = self.gbuf(p)
            acc = self.gacc(f1)
            if pidx is not None:
                self.bwd.add("ksmi_maxpool3x3s2_backward_idx", lambda: (pidx.data_ptr(), dp.data_ptr(), df1.data_ptr(), acc, B, H1, W1, 64, dt),
                             self._elt_meta("maxpool3_bwd", 2 * R1 * 64 + R1 * 64 // 2 + R1 * 64 // 4))
            else:
                self.bwd.add("ksmi_maxpool3x3s2_backward", lambda: (f1.data_ptr(), dp.data_ptr(), df1.data_ptr(), acc, B, H1, W1, 64, dt),
                             self._elt_meta("maxpool3_bwd", 4 * R1 * 64))
            self._bnrelu_bwd("resnet.bn1", df1, f1, s0, sv0, ds0, R1, 64)
            self._linear_bwd("resnet.conv1", col, Kpad, "resnet.conv1.weight", None, ds0, 64, R1, None, k_real=Kreal)
        self._bwd.append(bwd)
        return p, H2, W2

    def _absdiff(self, s1, s2, Cc, h, w):
        B, dt = self.B, self.dt
        n = B * h * w * Cc
        dbuf = self.buf(B, h, w, Cc)
        self.fwd.add("ksmi_absdiff_forward", lambda: (s1.data_ptr(), s2.data_ptr(), dbuf.data_ptr(), n, dt), self._elt_meta("absdiff", 3 * n))

        def bwd():
            dd, d1, d2 = self.gbuf(dbuf), self.gbuf(s1), self.gbuf(s2)
            a1, a2 = self.gacc(s1), self.gacc(s2)
            self.bwd.add("ksmi_absdiff_backward", lambda: (s1.data_ptr(), s2.data_ptr(), dd.data_ptr(), d1.data_ptr(), d2.data_ptr(), a1, a2, n, dt),
                         self._elt_meta("absdiff_bwd", 5 * n))
        self._bwd.append(bwd)
        return dbuf

    # ---------------------------------------------------------------- the graph
    def _backbone(self, date, stages=4, pred=None):
        """forward_single (bit_cd.py:780-797) of one date: stem, layer1..`stages`, nearest x2, conv_pred -> (prediction map, h, w).
        `pred`: destination of conv_pred (a view of a buffer both dates share), else a new buffer."""
        m, B, dt = self.m, self.B, self.dt
        t, h, w = self._stem(date)
        cin = 64
        for li, (ch, stride) in enumerate(LAYERS[:stages]):
            for bi in range(2):
                s_ = stride if bi == 0 else 1
                t = self._basic_block(f"resnet.layer{li + 1}.{bi}", t, cin, ch, h, w, s_)
                h, w, cin = h // s_, w // s_, ch
            self.named[f"layer{li + 1}_{date + 1}"] = t
        # nn.Upsample(scale_factor=2) (nearest) -> conv_pred
        U = self.buf(B, 2 * h, 2 * w, cin)
        self.fwd.add("ksmi_upsample2_forward", lambda t=t, U=U, h=h, w=w: (t.data_ptr(), U.data_ptr(), B, h, w, cin, 0, dt),
                     self._elt_meta("upsample2", 5 * B * h * w * cin))
        h2, w2 = 2 * h, 2 * w
        Pd = self.buf(B, h2, w2, 32) if pred is None else pred
        self._cv(self.fwd, "conv_pred", [SrcSpec(U, cin)], [(Pd, 32, 0, 0, 32, 0)], "conv_pred.weight", h2, w2, h2, w2, 3, 1, 1, 32, cin,
                 bias=m._p("conv_pred.bias"))
        self.named[f"pred_{date + 1}"] = Pd

        def bwd(t=t, U=U, Pd=Pd, h=h, w=w, h2=h2, w2=w2, cin=cin):
            dPd, dU = self.gbuf(Pd), self.buf(B, h2, w2, cin)
            self._wg([SrcSpec(U, cin)], dPd, 32, "conv_pred.weight", h2, w2, h2, w2, 3, 1, 1, cin)
            self._conv3(self.bwd, "conv_pred", [SrcSpec(dPd, 32)], [(dU, cin, 0, 0, cin, 0)], "conv_pred.weight", None, B, h2, w2, cin, 32, dgrad=True)
            self._bias_grad(dPd, B * h2 * w2, 32, "conv_pred.bias")
            dt_ = self.gbuf(t)
            if self.gacc(t):
                raise RuntimeError("unexpected second writer of a backbone output gradient")
            self.bwd.add("ksmi_upsample2_backward", lambda: (dU.data_ptr(), None, dt_.data_ptr(), B, h, w, cin, 0, dt),
                         self._elt_meta("upsample2_bwd", 5 * B * h * w * cin))
        self._bwd.append(bwd)
        return Pd, h2, w2

    def _build_bitcd(self):
        (f1, h2, w2), (f2, _, _) = self._backbone(0), self._backbone(1)
        self._head(f1, f2, h2, w2, ("resnet.fc.weight", "resnet.fc.bias"))

    def _head(self, f1, f2, h2, w2, unused_keys):
        """|f1 - f2| -> bilinear x4 -> classifier (bit_cd.py:766-771, 416-424)"""
        m, B, H, W, dt, nc = self.m, self.B, self.H, self.W, self.dt, self.nc
        D = self._absdiff(f1, f2, 32, h2, w2)
        X = self.buf(B, H, W, 32)
        self.fwd.add("ksmi_bilinear_forward", lambda: (D.data_ptr(), None, X.data_ptr(), B, h2, w2, H, W, 32, dt), self._elt_meta("bilinear", 2 * B * H * W * 32))

        def bil_bwd():
            dX, dD = self.gbuf(X), self.gbuf(D)
            self.gacc(D)
            self.bwd.add("ksmi_bilinear_backward", lambda: (dX.data_ptr(), dD.data_ptr(), 0, B, h2, w2, H, W, 32, dt), self._elt_meta("bilinear_bwd", 2 * B * H * W * 32))
        self._bwd.append(bil_bwd)
        # classifier: conv3x3 (no bias) -> BN -> ReLU -> conv3x3 (bias)
        npix = B * H * W
        z, y = self.buf(B, H, W, 32), self.buf(B, H, W, 32)
        sv = _Saved(32, self.dev)
        rows, cpad = self._cv(self.fwd, "classifier.0", [SrcSpec(X, 32)], [(z, 32, 0, 0, 32, 0)], "classifier.0.weight", H, W, H, W, 3, 1, 1, 32, 32,
                              stats=self.training)
        self._bn_finalize("classifier.1", sv, rows, cpad, 32, npix)
        self._affine(self.fwd, z, sv, y, npix, 32, 1)
        self.named["cls"] = y
        P = self.buf(B, H, W, CS)
        wk, bk = "classifier.3.weight", "classifier.3.bias"
        self._cv(self.fwd, "classifier.3", [SrcSpec(y, 32)], [(P, CS, 0, 0, nc, 0)], wk, H, W, H, W, 3, 1, 1, nc, 32, bias=m._p(bk))
        HW = H * W
        self.fwd.add("ksmi_out_to_nchw", lambda: (P.data_ptr(), self.logits.data_ptr(), B, nc, CS, HW, 0, dt))

        def head_bwd():
            dP = self.buf(B * HW, CS)
            dy = self.gbuf(y)
            self.gacc(y)
            self.bwd.add("ksmi_dout_to_nhwc", lambda: (self.dlogits.data_ptr(), self.logits.data_ptr(), dP.data_ptr(), B, nc, CS, HW, 0, dt))
            psrc = [SrcSpec(dP, CS, 0, CS, k_real=nc)]
            self._conv3(self.bwd, "classifier.3", psrc, [(dy, 32, 0, 0, 32, 0)], wk, None, B, H, W, 32, nc, dgrad=True)
            gview = m._g(wk)[8:]
            self.keep.append(gview)
            dw, ws = make_wgrad(psrc, y, 32, 0, 32, gview, 32 * 9, 9, -1, self._acc_param(wk), B, H, W, H, W, 3, 3, 1, 1, self.dtype)
            self._wgrad(dw, ws, wk)
            rr = max(1, min(512, B * HW // 256))
            self.need("red", rr * CS * 4)
            accb = self._acc_param(bk)
            gb = m._g(bk).data_ptr()
            self.bwd.add("ksmi_channel_sum", lambda: (dP.data_ptr(), self.scr("red"), rr, B * HW, CS, dt), self._elt_meta("channel_sum", B * HW * CS))
            self.bwd.add("ksmi_reduce_rows", lambda: (self.scr("red"), rr, 1, CS, nc, None, None, gb, accb))
            self._mark(bk)
            dz = self.buf(B, H, W, 32)
            self._bnrelu_bwd("classifier.1", dy, y, z, sv, dz, npix, 32)
            self._wg([SrcSpec(X, 32)], dz, 32, "classifier.0.weight", H, W, H, W, 3, 1, 1, 32)
            dX = self.gbuf(X)
            self._conv3(self.bwd, "classifier.0", [SrcSpec(dz, 32)], [(dX, 32, 0, 0, 32, self.gacc(X))], "classifier.0.weight", None, B, H, W, 32, 32, dgrad=True)
            for key in unused_keys:       # the unused ImageNet head of the backbone (and what resnet_stages_num cuts off): no gradient
                self._zero_grad_key(key)
        self._bwd.append(head_bwd)

    # ---------------------------------------------------------------- execution
    def run_forward(self, x1, x2):
        if x1.data_ptr() != self.x[0].data_ptr():
            self.x[0].copy_(x1)
        if x2.data_ptr() != self.x[1].data_ptr():
            self.x[1].copy_(x2)
        self.packs.run()
        self.fwd.run()
        return self.logits


class BitCDTransformerPlan(BitCDPlan):
    """`BASE_Transformer` (bit_cd.py:802-934; the three transformer variants of define_G :690-700): ResNet-18 cut after layer3, semantic
    tokenizer, token encoder, per-date token decoder, then the head of the base network.

    Pixel side ([2B * h * w][32] rows in the plan's dtype, both dates stacked date-major): ksmi_semantic_tokens_*, ksmi_token_cross_*
    (csrc/bitcd.hip) and the LayerNorm / Linear / GELU launches of the token rows of the FloodViT path for the decoder's feed-forward.
    Token side (2 * token_len tokens per image pair, fp32): LayerNorm + strided batched products (ksmi_bmm_f32) + row softmax."""

    F32 = 0        # KSMI_F32

    def _build_bitcd(self):
        m, B = self.m, self.B
        h2, w2 = self.H // 4, self.W // 4                    # stem /4, layer2 /2, layer3 stride 1, nearest x2
        PP = self.buf(2 * B, h2, w2, 32)
        P1, P2 = PP[:B], PP[B:]
        self.keep += [P1, P2]
        (f1, _, _), (f2, _, _) = self._backbone(0, 3, P1), self._backbone(1, 3, P2)
        N = h2 * w2
        Y = self._tokens(PP, N)
        Y1, Y2 = Y[:B], Y[B:]
        gY = self.buf(2 * B, h2, w2, 32) if self.with_backward else None      # ONE gradient buffer: d|y1 - y2| lands here, every decoder
        if self.with_backward:                                               # layer updates it in place, the tokenizer adds to it,
            self._gbuf[id(Y1)], self._gbuf[id(Y2)] = gY[:B], gY[B:]          # conv_pred's backward reads it per date
            self._gbuf[id(P1)], self._gbuf[id(P2)] = gY[:B], gY[B:]
            self._gx = gY
            self.keep += [Y1, Y2]
        self.named["dec_1"], self.named["dec_2"] = Y1, Y2
        unused = [k for k in m._pspec if k.startswith(("resnet.layer4.", "resnet.fc."))]
        self._head(Y1, Y2, h2, w2, unused)

    # ---------------------------------------------------------------- small fp32 helpers (token side)
    def _st(self, *v):
        a = (C.c_int64 * 4)(*v)
        self.keep.append(a)
        return a

    def _bmm(self, ll, a, b, c, nb1, nb2, M, N, K, sa, sb, sc, bias=None, alpha=1.0, acc=0, tag="bmm"):
        """a, b, c: data pointers (ints); strides as in ksmi_bmm_f32"""
        sa, sb, sc = self._st(*sa), self._st(*sb), self._st(*sc)
        ll.add("ksmi_bmm_f32", lambda: (a, b, bias, c, nb1, nb2, M, N, K, sa, sb, sc, alpha, acc),
               {"kind": "bmm_f32", "bytes": 4 * nb1 * nb2 * (M * K + K * N + M * N), "flops": 2 * nb1 * nb2 * M * N * K, "tag": tag})

    def _lin32(self, name, x, rows, Cin, wkey, bkey, out, N):
        """out[rows][N] = x[rows][Cin] W^T + b (fp32 token rows)"""
        bp = self.m._p(bkey).data_ptr() if bkey else None
        self._bmm(self.fwd, x.data_ptr(), self.m._p(wkey).data_ptr(), out.data_ptr(), 1, 1, rows, N, Cin, (0, 0, Cin, 1), (0, 0, 1, Cin), (0, 0, N, 1),
                  bias=bp, tag=name)

    def _lin32_bwd(self, name, x, rows, Cin, wkey, bkey, dy, N, dx, dx_acc=0):
        """dx (+)= dy W ; dW (+)= dy^T x ; db (+)= colsum(dy)"""
        w, gw = self.m._p(wkey).data_ptr(), self.m._g(wkey).data_ptr()
        if dx is not None:
            self._bmm(self.bwd, dy.data_ptr(), w, dx.data_ptr(), 1, 1, rows, Cin, N, (0, 0, N, 1), (0, 0, Cin, 1), (0, 0, Cin, 1), acc=dx_acc, tag=f"{name}.dx")
        self._bmm(self.bwd, dy.data_ptr(), x.data_ptr(), gw, 1, 1, N, Cin, rows, (0, 0, 1, N), (0, 0, Cin, 1), (0, 0, Cin, 1), acc=self._acc_param(wkey),
                  tag=f"{name}.dW")
        self._mark(wkey)
        if bkey:
            acc = self._acc_param(bkey)
            gb = self.m._g(bkey).data_ptr()
            self.bwd.add("ksmi_colsum", lambda: (dy.data_ptr(), rows, N, gb, acc, self.F32), {"kind": "colsum", "bytes": 4 * rows * N, "flops": 0})
            self._mark(bkey)

    def _ln32(self, x, wkey, bkey, y, rows):
        st = self.fbuf(2, rows)
        g, b = self.m._p(wkey).data_ptr(), self.m._p(bkey).data_ptr()
        self.fwd.add("ksmi_layernorm_forward", lambda: (x.data_ptr(), g, b, y.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), rows, 32, 1e-5, self.F32),
                     {"kind": "layernorm_fwd", "bytes": 8 * rows * 32, "flops": 0})
        return st

    def _ln32_bwd(self, dy, x, st, wkey, bkey, dx, accumulate, rows):
        nblk = self.lib.ksmi_layernorm_bwd_blocks(rows)
        self.need("lnp", nblk * 2 * 32 * 4)
        g = self.m._p(wkey).data_ptr()
        self.bwd.add("ksmi_layernorm_backward", lambda: (dy.data_ptr(), x.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), g, dx.data_ptr(), accumulate,
                                                         self.scr("lnp"), rows, 32, self.F32), {"kind": "layernorm_bwd", "bytes": 16 * rows * 32, "flops": 0})
        a1, a2 = self._acc_param(wkey), self._acc_param(bkey)
        assert a1 == a2
        gw, gb = self.m._g(wkey).data_ptr(), self.m._g(bkey).data_ptr()
        self.bwd.add("ksmi_reduce_rows", lambda: (self.scr("lnp"), nblk, 2, 32, 32, None, gw, gb, a1))
        self._mark(wkey, bkey)

    # ---------------------------------------------------------------- tokenizer -> encoder -> decoder
    def _tokens(self, PP, N):
        m, B, dt, F32 = self.m, self.B, self.dt, self.F32
        L, H = m.token_len, 8
        R8, Rp = B * 2 * L, 2 * B * N                       # token rows, pixel rows
        scale = 32 ** -0.5                                  # Attention / Cross_Attention: dim ** -0.5 with dim = 32 (bit_cd.py:481,531)
        wb = self.with_backward
        steps = []                                          # backward closures of this stage, forward order
        P, G = (lambda k: m._p(k).data_ptr()), (lambda k: m._g(k).data_ptr())

        # ---- semantic tokens of both dates + position table (:857-865, :880-881)
        T0, tstats = self.fbuf(B, 2 * L, 32), self.fbuf(2 * B, L, 2)
        pos = P("pos_embedding") if m.with_pos else None
        self.fwd.add("ksmi_semantic_tokens_forward", lambda: (PP.data_ptr(), P("conv_a.weight"), pos, T0.data_ptr(), tstats.data_ptr(), B, 2, N, 32, L, dt),
                     {"kind": "semantic_tokens", "bytes": 2 * Rp * 32 * self._es(), "flops": 4 * Rp * 32 * L})
        gT = self.fbuf(B, 2 * L, 32) if wb else None         # gradient of the token residual stream (encoder output ... encoder input)

        def tokenizer_bwd():
            if m.with_pos:
                gp = G("pos_embedding")
                acc = self._acc_param("pos_embedding")
                self.bwd.add("ksmi_batch_sum", lambda: (gT.data_ptr(), gp, B, 2 * L * 32, acc, F32), {"kind": "batch_sum", "bytes": 4 * R8 * 32, "flops": 0})
                self._mark("pos_embedding")
            part = self.fbuf(2 * B, L * 32)
            gx = self._gx
            self.bwd.add("ksmi_semantic_tokens_backward", lambda: (PP.data_ptr(), P("conv_a.weight"), tstats.data_ptr(), gT.data_ptr(), gx.data_ptr(),
                                                                   part.data_ptr(), B, 2, N, 32, L, 1, dt),
                         {"kind": "semantic_tokens_bwd", "bytes": 4 * Rp * 32 * self._es(), "flops": 10 * Rp * 32 * L})
            acc = self._acc_param("conv_a.weight")
            ga = G("conv_a.weight")
            self.bwd.add("ksmi_reduce_rows", lambda: (part.data_ptr(), 2 * B, 1, L * 32, L * 32, None, None, ga, acc))
            self._mark("conv_a.weight")
        steps.append(tokenizer_bwd)

        # ---- token encoder (:564-578, Attention :527-561): fp32, 2L tokens per pair
        T = T0
        n, Dh = 2 * L, m.dim_head
        I = H * Dh
        for li in range(m.enc_depth):
            a, f = f"transformer.layers.{li}.0.fn", f"transformer.layers.{li}.1.fn"
            x_in = T
            h1, qkv, dots, attn, att = self.fbuf(R8, 32), self.fbuf(R8, 3 * I), self.fbuf(B, H, n, n), self.fbuf(B, H, n, n), self.fbuf(R8, I)
            tmp, x_mid, h2_, u, g, x_out = self.fbuf(R8, 32), self.fbuf(R8, 32), self.fbuf(R8, 32), self.fbuf(R8, 64), self.fbuf(R8, 64), self.fbuf(R8, 32)
            st1 = self._ln32(x_in, f"{a}.norm.weight", f"{a}.norm.bias", h1, R8)
            self._lin32(f"enc{li}.to_qkv", h1, R8, 32, f"{a}.fn.to_qkv.weight", None, qkv, 3 * I)
            q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * I, qkv.data_ptr() + 8 * I
            rs = 3 * I
            self._bmm(self.fwd, q, k, dots.data_ptr(), B, H, n, n, Dh, (n * rs, Dh, rs, 1), (n * rs, Dh, 1, rs), (H * n * n, n * n, n, 1), tag=f"enc{li}.qk")
            self.fwd.add("ksmi_softmax_rows_f32", lambda dots=dots, attn=attn: (dots.data_ptr(), attn.data_ptr(), B * H * n, n, scale))
            self._bmm(self.fwd, attn.data_ptr(), v, att.data_ptr(), B, H, n, Dh, n, (H * n * n, n * n, n, 1), (n * rs, Dh, rs, 1), (n * I, Dh, I, 1), tag=f"enc{li}.pv")
            self._lin32(f"enc{li}.to_out", att, R8, I, f"{a}.fn.to_out.0.weight", f"{a}.fn.to_out.0.bias", tmp, 32)
            self.fwd.add("ksmi_add", lambda tmp=tmp, x_in=x_in, x_mid=x_mid: (tmp.data_ptr(), x_in.data_ptr(), x_mid.data_ptr(), R8 * 32, F32))
            st2 = self._ln32(x_mid, f"{f}.norm.weight", f"{f}.norm.bias", h2_, R8)
            self._lin32(f"enc{li}.ff1", h2_, R8, 32, f"{f}.fn.net.0.weight", f"{f}.fn.net.0.bias", u, 64)
            self.fwd.add("ksmi_gelu_forward", lambda u=u, g=g: (u.data_ptr(), g.data_ptr(), R8 * 64, F32))
            self._lin32(f"enc{li}.ff2", g, R8, 64, f"{f}.fn.net.3.weight", f"{f}.fn.net.3.bias", tmp, 32)
            self.fwd.add("ksmi_add", lambda tmp=tmp, x_mid=x_mid, x_out=x_out: (tmp.data_ptr(), x_mid.data_ptr(), x_out.data_ptr(), R8 * 32, F32))
            T = x_out

            def enc_bwd(li=li, a=a, f=f, x_in=x_in, h1=h1, qkv=qkv, attn=attn, att=att, x_mid=x_mid, h2_=h2_, u=u, g=g, st1=st1, st2=st2):
                tM, tD, tI, dqkv, dat, ds = self.fbuf(R8, 64), self.fbuf(R8, 32), self.fbuf(R8, I), self.fbuf(R8, 3 * I), self.fbuf(B, H, n, n), self.fbuf(B, H, n, n)
                q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * I, qkv.data_ptr() + 8 * I
                dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + 4 * I, dqkv.data_ptr() + 8 * I
                self._lin32_bwd(f"enc{li}.ff2", g, R8, 64, f"{f}.fn.net.3.weight", f"{f}.fn.net.3.bias", gT, 32, tM)
                self.bwd.add("ksmi_gelu_backward", lambda: (tM.data_ptr(), u.data_ptr(), tM.data_ptr(), R8 * 64, F32))
                self._lin32_bwd(f"enc{li}.ff1", h2_, R8, 32, f"{f}.fn.net.0.weight", f"{f}.fn.net.0.bias", tM, 64, tD)
                self._ln32_bwd(tD, x_mid, st2, f"{f}.norm.weight", f"{f}.norm.bias", gT, 1, R8)
                self._lin32_bwd(f"enc{li}.to_out", att, R8, I, f"{a}.fn.to_out.0.weight", f"{a}.fn.to_out.0.bias", gT, 32, tI)
                ti = tI.data_ptr()
                self._bmm(self.bwd, ti, v, dat.data_ptr(), B, H, n, n, Dh, (n * I, Dh, I, 1), (n * rs, Dh, 1, rs), (H * n * n, n * n, n, 1), tag=f"enc{li}.dP")
                self._bmm(self.bwd, attn.data_ptr(), ti, dv, B, H, n, Dh, n, (H * n * n, n * n, 1, n), (n * I, Dh, I, 1), (n * rs, Dh, rs, 1), tag=f"enc{li}.dV")
                self.bwd.add("ksmi_softmax_rows_backward_f32", lambda: (attn.data_ptr(), dat.data_ptr(), ds.data_ptr(), B * H * n, n, scale))
                self._bmm(self.bwd, ds.data_ptr(), k, dq, B, H, n, Dh, n, (H * n * n, n * n, n, 1), (n * rs, Dh, rs, 1), (n * rs, Dh, rs, 1), tag=f"enc{li}.dQ")
                self._bmm(self.bwd, ds.data_ptr(), q, dk, B, H, n, Dh, n, (H * n * n, n * n, 1, n), (n * rs, Dh, rs, 1), (n * rs, Dh, rs, 1), tag=f"enc{li}.dK")
                self._lin32_bwd(f"enc{li}.to_qkv", h1, R8, 32, f"{a}.fn.to_qkv.weight", None, dqkv, 3 * I, tD)
                self._ln32_bwd(tD, x_in, st1, f"{a}.norm.weight", f"{a}.norm.bias", gT, 1, R8)
            steps.append(enc_bwd)
        self.named["tokens"] = T

        # ---- token decoder on the pixels of both dates (:581-598; Cross_Attention :476-524 behind PreNorm2 / Residual2)
        Dd = m.decoder_dim_head
        Id = H * Dd
        X = PP.view(Rp, 32)
        if wb:
            tD_, tM_ = self.buf(Rp, 32), self.buf(Rp, 64)
            self.need("cross", self.lib.ksmi_token_cross_bwd_workspace(B, 2, N))
        mem_steps = []
        # token side of ALL decoder layers at once: every layer reads the same encoder output T, so k, v and the folded matrices of
        # the dec_depth layers are batched products over a layer axis (the layers' parameters sit at one constant stride in the arena:
        # same keys, same order), ahead of the pixel loop; the backward unfolds the layers' dA / dBv the same way behind it
        Ld = m.dec_depth
        dkey = lambda li, sfx: f"transformer_decoder.layers.{li}.0.fn.{sfx}"
        lstride = (m._poff[dkey(1, "fn.to_q.weight")] - m._poff[dkey(0, "fn.to_q.weight")]) if Ld > 1 else 0
        for sfx in ("norm.weight", "fn.to_q.weight", "fn.to_k.weight", "fn.to_v.weight", "fn.to_out.0.weight", "fn.to_out.0.bias"):
            for li in range(1, Ld):
                assert m._poff[dkey(li, sfx)] - m._poff[dkey(li - 1, sfx)] == lstride, "decoder layers are not at a constant arena stride"
        W0 = lambda sfx: m._p(dkey(0, sfx)).data_ptr()
        G0 = lambda sfx: m._g(dkey(0, sfx)).data_ptr()
        mn_all, K_all, V_all = self.fbuf(Ld, R8, 32), self.fbuf(Ld, R8, Id), self.fbuf(Ld, R8, Id)
        A_all, Bv_all = self.fbuf(Ld, R8, H, 32), self.fbuf(Ld, R8, H, 32)
        stms = [self._ln32(T, dkey(li, "norm.weight"), dkey(li, "norm.bias"), mn_all[li], R8) for li in range(Ld)]
        self._bmm(self.fwd, mn_all.data_ptr(), W0("fn.to_k.weight"), K_all.data_ptr(), Ld, 1, R8, Id, 32, (R8 * 32, 0, 32, 1), (lstride, 0, 1, 32),
                  (R8 * Id, 0, Id, 1), tag="dec.to_k")
        self._bmm(self.fwd, mn_all.data_ptr(), W0("fn.to_v.weight"), V_all.data_ptr(), Ld, 1, R8, Id, 32, (R8 * 32, 0, 32, 1), (lstride, 0, 1, 32),
                  (R8 * Id, 0, Id, 1), tag="dec.to_v")
        self._bmm(self.fwd, K_all.data_ptr(), W0("fn.to_q.weight"), A_all.data_ptr(), Ld, H, R8, 32, Dd, (R8 * Id, Dd, Id, 1), (lstride, Dd * 32, 32, 1),
                  (R8 * H * 32, 32, H * 32, 1), tag="dec.A")
        self._bmm(self.fwd, V_all.data_ptr(), W0("fn.to_out.0.weight"), Bv_all.data_ptr(), Ld, H, R8, 32, Dd, (R8 * Id, Dd, Id, 1), (lstride, Dd, 1, Id),
                  (R8 * H * 32, 32, H * 32, 1), tag="dec.Bv")
        dA_all = self.fbuf(Ld, R8, H, 32) if wb else None
        dBv_all = self.fbuf(Ld, R8, H, 32) if wb else None
        for li in range(Ld):
            a, f = f"transformer_decoder.layers.{li}.0.fn", f"transformer_decoder.layers.{li}.1.fn"
            nk, bo = f"{a}.norm", f"{a}.fn.to_out.0.bias"
            A_, Bv = A_all[li], Bv_all[li]
            # pixel side
            x_in, x_mid, h2_, u, g, x_out = X, self.buf(Rp, 32), self.buf(Rp, 32), self.buf(Rp, 64), self.buf(Rp, 64), self.buf(Rp, 32)
            self.fwd.add("ksmi_token_cross_forward", lambda x_in=x_in, x_mid=x_mid, A_=A_, Bv=Bv, nk=nk, bo=bo: (
                x_in.data_ptr(), P(f"{nk}.weight"), P(f"{nk}.bias"), A_.data_ptr(), Bv.data_ptr(), P(bo), x_mid.data_ptr(), B, 2, N, 32, H, L, scale, dt),
                {"kind": "token_cross_fwd", "bytes": 2 * Rp * 32 * self._es(), "flops": 4 * Rp * 32 * 32})
            st2 = self._ln(x_mid, f"{f}.norm.weight", f"{f}.norm.bias", h2_, Rp, 32, eps=1e-5)
            self._linear(f"dec{li}.ff1", h2_, 32, f"{f}.fn.net.0.weight", f"{f}.fn.net.0.bias", u, 64, Rp)
            self.fwd.add("ksmi_gelu_forward", lambda u=u, g=g: (u.data_ptr(), g.data_ptr(), Rp * 64, dt), self._elt_meta("gelu", 2 * Rp * 64))
            self._linear(f"dec{li}.ff2", g, 64, f"{f}.fn.net.3.weight", f"{f}.fn.net.3.bias", x_out, 32, Rp, resid=x_mid)
            X = x_out

            def dec_bwd(li=li, f=f, nk=nk, bo=bo, A_=A_, Bv=Bv, x_in=x_in, x_mid=x_mid, h2_=h2_, u=u, g=g, st2=st2):
                gx = self._gx.view(Rp, 32)
                # feed-forward on the pixels: x_out = x_mid + W2 gelu(W1 LN(x_mid) + b1) + b2
                self._linear_bwd(f"dec{li}.ff2", g, 64, f"{f}.fn.net.3.weight", f"{f}.fn.net.3.bias", gx, 32, Rp, tM_)
                self.bwd.add("ksmi_gelu_backward", lambda: (tM_.data_ptr(), u.data_ptr(), tM_.data_ptr(), Rp * 64, dt), self._elt_meta("gelu_bwd", 3 * Rp * 64))
                self._linear_bwd(f"dec{li}.ff1", h2_, 32, f"{f}.fn.net.0.weight", f"{f}.fn.net.0.bias", tM_, 64, Rp, tD_)
                self._ln_bwd(tD_, x_mid, st2, f"{f}.norm.weight", f"{f}.norm.bias", gx, 1, Rp, 32)
                # cross-attention: pixels (gx in place, LayerNorm / to_out bias gradients) and the layer's folded-matrix gradients
                dA, dBv = dA_all[li], dBv_all[li]
                a_ln, a_bo = self._acc_param(f"{nk}.weight"), self._acc_param(bo)
                assert self._acc_param(f"{nk}.bias") == a_ln
                self.bwd.add("ksmi_token_cross_backward", lambda: (
                    x_in.data_ptr(), P(f"{nk}.weight"), P(f"{nk}.bias"), A_.data_ptr(), Bv.data_ptr(), gx.data_ptr(), dA.data_ptr(), dBv.data_ptr(),
                    G(f"{nk}.weight"), G(f"{nk}.bias"), G(bo), a_ln, a_bo, self.scr("cross"), B, 2, N, 32, H, L, scale, dt),
                    {"kind": "token_cross_bwd", "bytes": 3 * Rp * 32 * self._es(), "flops": 12 * Rp * 32 * 32})
                self._mark(bo)
            mem_steps.append(dec_bwd)

        def dec_tokens_bwd():
            """behind the pixel loop: unfold dA -> (dWq, dK), dBv -> (dWo, dV), to_k / to_v, for all layers at once; then the LayerNorm of
            the tokens layer by layer (its parameters are the ones the pixels use: the pixel kernel wrote their gradients, these add)"""
            dK_all, dV_all, dmn_all = self.fbuf(Ld, R8, Id), self.fbuf(Ld, R8, Id), self.fbuf(Ld, R8, 32)
            accs = {sfx: {self._acc_param(dkey(li, sfx)) for li in range(Ld)} for sfx in ("fn.to_q.weight", "fn.to_k.weight", "fn.to_v.weight", "fn.to_out.0.weight")}
            assert all(len(v) == 1 for v in accs.values())
            acc = {k: v.pop() for k, v in accs.items()}
            da, dbv, dk, dv, dmn = dA_all.data_ptr(), dBv_all.data_ptr(), dK_all.data_ptr(), dV_all.data_ptr(), dmn_all.data_ptr()
            sA = (R8 * H * 32, 32, H * 32, 1)
            self._bmm(self.bwd, da, W0("fn.to_q.weight"), dk, Ld, H, R8, Dd, 32, sA, (lstride, Dd * 32, 1, 32), (R8 * Id, Dd, Id, 1), tag="dec.dK")
            self._bmm(self.bwd, K_all.data_ptr(), da, G0("fn.to_q.weight"), Ld, H, Dd, 32, R8, (R8 * Id, Dd, 1, Id), sA, (lstride, Dd * 32, 32, 1),
                      acc=acc["fn.to_q.weight"], tag="dec.dWq")
            self._bmm(self.bwd, dbv, W0("fn.to_out.0.weight"), dv, Ld, H, R8, Dd, 32, sA, (lstride, Dd, Id, 1), (R8 * Id, Dd, Id, 1), tag="dec.dV")
            self._bmm(self.bwd, dbv, V_all.data_ptr(), G0("fn.to_out.0.weight"), Ld, H, 32, Dd, R8, (R8 * H * 32, 32, 1, H * 32), (R8 * Id, Dd, Id, 1),
                      (lstride, Dd, Id, 1), acc=acc["fn.to_out.0.weight"], tag="dec.dWo")
            # to_k / to_v: d LN(T) = dK Wk + dV Wv ; dWk = dK^T LN(T) ; dWv = dV^T LN(T)
            sM, sKV = (R8 * 32, 0, 32, 1), (R8 * Id, 0, Id, 1)
            self._bmm(self.bwd, dk, W0("fn.to_k.weight"), dmn, Ld, 1, R8, 32, Id, sKV, (lstride, 0, 32, 1), sM, tag="dec.to_k.dx")
            self._bmm(self.bwd, dv, W0("fn.to_v.weight"), dmn, Ld, 1, R8, 32, Id, sKV, (lstride, 0, 32, 1), sM, acc=1, tag="dec.to_v.dx")
            self._bmm(self.bwd, dk, mn_all.data_ptr(), G0("fn.to_k.weight"), Ld, 1, Id, 32, R8, (R8 * Id, 0, 1, Id), sM, (lstride, 0, 32, 1),
                      acc=acc["fn.to_k.weight"], tag="dec.to_k.dW")
            self._bmm(self.bwd, dv, mn_all.data_ptr(), G0("fn.to_v.weight"), Ld, 1, Id, 32, R8, (R8 * Id, 0, 1, Id), sM, (lstride, 0, 32, 1),
                      acc=acc["fn.to_v.weight"], tag="dec.to_v.dW")
            for li in range(Ld):
                self._mark(*[dkey(li, sfx) for sfx in ("fn.to_q.weight", "fn.to_k.weight", "fn.to_v.weight", "fn.to_out.0.weight")])
            for li in reversed(range(Ld)):
                self._ln32_bwd(dmn_all[li], T, stms[li], dkey(li, "norm.weight"), dkey(li, "norm.bias"), gT, self._gT_started(), R8)
        self._gT_acc = 0
        # backward order: pixel side of the decoder layers (last first), their token side, encoder (last first), tokenizer
        if wb:
            def token_stage_bwd():
                for fn in reversed(mem_steps):
                    fn()
                dec_tokens_bwd()
                for fn in reversed(steps):
                    fn()
            self._bwd.append(token_stage_bwd)
        return X.view(2 * B, PP.shape[1], PP.shape[2], 32)

    def _gT_started(self):
        """accumulate flag of the token gradient: the first decoder layer of the backward pass writes it, the others add"""
        a = self._gT_acc
        self._gT_acc = 1
        return a
