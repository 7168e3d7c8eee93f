"""Static launch plan of one FloodViT forward/backward at a fixed (B, dtype).

Token activations are [rows = B*197][C] in the activation dtype (an NHWC image of rows x 1 pixels), so every
nn.Linear is a 1x1 implicit GEMM on ksmi_conv_forward / ksmi_conv_wgrad; the Decoder's ConvTranspose2d(k4,s2,p1)
layers run as four 2x2 phase convolutions (forward), one 4x4 stride-2 convolution (input gradient) and one 4x4
stride-2 weight-gradient GEMM.

Reference computation: /root/reference/models/vision_transformer.py:139-153 (ViT.forward), :35-66 (Attention),
:19-32 (FeedForward), :69-89 (Transformer); /root/reference/models/model_utilities.py:80-94
(FinetunerSegmentation.forward), :36-48 (Decoder.forward), :59-72 (the `mlp` and default 1x1 heads).
"""
import os

import torch

from . import _lib
from .plan_base import PlanBase
from .runtime import SrcSpec, make_conv, make_wgrad


class FloodViTPlan(PlanBase):
    # the nn.Linear weight gradients of the transformer layers on the train step's side stream (plan_base.PlanBase.side_tokens)
    side_tokens = os.environ.get("KSMI_SIDE_TOKENS", "1") != "0"
    input_names = ("x",)

    def __init__(self, model, B, dtype, with_backward):
        self._init_base(model, dtype, with_backward)
        self.B = B
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
        self._finish()

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

        gx = self.buf(R, D) if self.with_backward else None     # gradient of the residual stream, updated in place layer by layer
        tD = self.buf(R, D) if self.with_backward else None

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
        X = self._transformer_layers(X, self.depth, "model.transformer", B, self.N1, D, self.heads, self.m.hp["dim_head"], M, gx, bwd_steps)

        # ---- final LN, drop cls, Decoder (vision_transformer.py:89,150-151; model_utilities.py:85-93,36-48) ----
        XF, F = self.buf(R, D), self.buf(Rp, D)
        x_last = X
        st_f = self._ln(x_last, "model.transformer.norm.weight", "model.transformer.norm.bias", XF, R, D)
        self.fwd.add("ksmi_drop_cls", lambda: (XF.data_ptr(), F.data_ptr(), B, self.N1, D, 0, dt), self._elt_meta("drop_cls", 2 * Rp * D))
        gh, gw = self.gh, self.gw
        HW = self.ih * self.iw
        L = self.buf(B, self.ih, self.iw, self.Cs)
        dL = self.buf(B, self.ih, self.iw, self.Cs) if self.with_backward else None
        dF = self.buf(Rp, D) if self.with_backward else None
        kind = self.m.head_kind
        head_bwd = {"decoder": self._head_decoder, "mlp": self._head_mlp, "linear": self._head_linear}[kind](F, L, dL, dF)
        self.named.update(xf=XF, feat=F, logits_nhwc=L)
        self.fwd.add("ksmi_logits_to_nchw", lambda: (L.data_ptr(), self.logits.data_ptr(), B, self.ncls, self.Cs, HW, dt),
                     self._elt_meta("logits_to_nchw", B * HW * (self.ncls + 2 * self.ncls)))

        if not self.with_backward:
            return
        # ---- backward ---------------------------------------------------------------------------
        self.bwd.add("ksmi_dlogits_to_nhwc", lambda: (self.dlogits.data_ptr(), dL.data_ptr(), B, self.ncls, self.Cs, HW, dt),
                     self._elt_meta("dlogits_to_nhwc", B * HW * (2 * self.ncls + self.Cs)))
        head_bwd()
        if not self.train_encoder:
            return
        self.bwd.add("ksmi_drop_cls", lambda: (dF.data_ptr(), tD.data_ptr(), B, self.N1, D, 1, dt), self._elt_meta("drop_cls_bwd", 2 * R * D))
        self._ln_bwd(tD, x_last, st_f, "model.transformer.norm.weight", "model.transformer.norm.bias", gx, 0, R, D)
        for step in reversed(bwd_steps):
            step()

    # ---------------------------------------------------------------- heads of FinetunerSegmentation (model_utilities.py:59-72,85-93)
    def _head_decoder(self, F, L, dL, dF):
        """`decoder`: ConvT(1024->128) ReLU nearest x2 ConvT(128->64) ReLU ConvT(64->classes)  (model_utilities.py:22-48)"""
        B, D, dt, gh, gw = self.B, self.D, self.dt, self.gh, self.gw
        if (16 * gh, 16 * gw) != (self.ih, self.iw):
            raise _lib.KsmiError("FloodViT decoder expects patch size 16 (14 -> 28 -> 56 -> 112 -> 224, model_utilities.py:36-48)")
        D1 = self.buf(B, 2 * gh, 2 * gw, 128)
        U1 = self.buf(B, 4 * gh, 4 * gw, 128)
        D2 = self.buf(B, 8 * gh, 8 * gw, 64)
        self.named.update(d1=D1, u1=U1, d2=D2)
        self._deconv("deconv1", F, D, 128, gh, gw, D1, 128)
        self.fwd.add("ksmi_upsample2_forward", lambda: (D1.data_ptr(), U1.data_ptr(), B, 2 * gh, 2 * gw, 128, 1, dt),
                     self._elt_meta("upsample2", 5 * D1.numel()))
        self._deconv("deconv2", U1, 128, 64, 4 * gh, 4 * gw, D2, 64)
        self.fwd.add("ksmi_relu_forward", lambda: (D2.data_ptr(), D2.data_ptr(), D2.numel(), dt), self._elt_meta("relu", 2 * D2.numel()))
        self._deconv("deconv3", D2, 64, self.ncls, 8 * gh, 8 * gw, L, self.Cs)

        def bwd():
            dD2, dU1, dD1 = self.buf(*D2.shape), self.buf(*U1.shape), self.buf(*D1.shape)
            self._deconv_bwd("deconv3", D2, 64, self.ncls, 8 * gh, 8 * gw, dL, self.Cs, dD2, mask=D2)
            self._deconv_bwd("deconv2", U1, 128, 64, 4 * gh, 4 * gw, dD2, 64, dU1)
            self.bwd.add("ksmi_upsample2_backward", lambda: (dU1.data_ptr(), D1.data_ptr(), dD1.data_ptr(), B, 2 * gh, 2 * gw, 128, 1, dt),
                         self._elt_meta("upsample2_bwd", 6 * D1.numel()))
            self._deconv_bwd("deconv1", F, D, 128, gh, gw, dD1, 128, dF if self.train_encoder else None)
        return bwd

    def _conv1x1(self, name, x, Cin, wkey, bkey, out, outC, N, rows):
        """out[rows, outC][:, :N] = x[rows, Cin] @ W[N, Cin]^T + b   (nn.Conv2d k=1 on NHWC rows; outC >= N: padded channel stride)"""
        d, table = make_conv([SrcSpec(x, Cin)], [(out, outC, 0, 0, N, 0)], out, self.m._p(bkey), None, 1, rows, 1, rows, 1, 1, 1, 1, 0, N, self.dtype)
        d.wpk = self._packed(wkey, table, 1, N, N, 1, Cin, 0, 0).data_ptr()
        self._conv(self.fwd, d, "conv1x1", name)

    def _conv1x1_bwd(self, name, x, Cin, wkey, bkey, dy, dyC, N, rows, dx, mask=None):
        """dx = (dy @ W) [* (mask > 0)] ; dW = dy^T x ; db = column sums of the N real channels of dy"""
        src = [SrcSpec(dy, dyC, 0, dyC, k_real=N)]
        if dx is not None:
            mk = None if mask is None else (mask, self.const[0], self.const[1], self.const[1], self.const[0])
            d, table = make_conv(src, [(dx, Cin, 0, 0, Cin, 0)], dx, None, None, 1, rows, 1, rows, 1, 1, 1, 1, 0, Cin, self.dtype, mask=mk)
            d.wpk = self._packed(wkey, table, 1, Cin, Cin, Cin, 1, 0, 0).data_ptr()
            self._conv(self.bwd, d, "conv1x1_dgrad", name)
        dw, ws = make_wgrad([SrcSpec(x, Cin)], dy, dyC, 0, N, self.m._g(wkey), 1, Cin, 0, self._acc_param(wkey), 1, rows, 1, rows, 1, 1, 1, 1, 0,
                            self.dtype)
        self._wgrad(dw, ws, wkey)
        r = max(1, min(512, rows // 256))
        slot = self._rs_slot(r * dyC * 4)
        self.bwd.add("ksmi_channel_sum", lambda: (dy.data_ptr(), self.scr(slot), r, rows, dyC, self.dt), self._elt_meta("channel_sum", rows * dyC))
        self._defer_rowsum(bkey, slot, 0, r, 1, 0, dyC, N)
        self._rs_tick()

    def _head_linear(self, F, L, dL, dF):
        """default head: nn.Upsample(224^2, bilinear) then Conv1x1(1024 -> classes) (model_utilities.py:70-72,88-93).  A 1x1 convolution
        (bias included: the bilinear weights of a pixel sum to 1) commutes with the per-channel interpolation, so the convolution
        runs on the 14 x 14 map and the classes are interpolated: 256 x fewer MACs, same function."""
        B, D, dt, gh, gw = self.B, self.D, self.dt, self.gh, self.gw
        Y = self.buf(B, gh, gw, self.Cs)
        self._conv1x1("head", F, D, "head.weight", "head.bias", Y, self.Cs, self.ncls, self.Rp)
        self.fwd.add("ksmi_bilinear_forward", lambda: (Y.data_ptr(), None, L.data_ptr(), B, gh, gw, self.ih, self.iw, self.Cs, dt),
                     self._elt_meta("bilinear", 2 * L.numel()))

        def bwd():
            dY = self.buf(*Y.shape)
            self.bwd.add("ksmi_bilinear_backward", lambda: (dL.data_ptr(), dY.data_ptr(), 0, B, gh, gw, self.ih, self.iw, self.Cs, dt),
                         self._elt_meta("bilinear_bwd", 2 * dL.numel()))
            self._conv1x1_bwd("head", F, D, "head.weight", "head.bias", dY, self.Cs, self.ncls, self.Rp, dF if self.train_encoder else None)
        return bwd

    def _head_mlp(self, F, L, dL, dF):
        """`mlp` head: bilinear to 224^2, Conv1x1(1024 -> 512), ReLU, Conv1x1(512 -> classes) (model_utilities.py:59-64,88-93); the first
        convolution commutes with the interpolation (see _head_linear) and runs on the 14 x 14 map, the ReLU does not."""
        B, D, dt, gh, gw = self.B, self.D, self.dt, self.gh, self.gw
        rows = B * self.ih * self.iw
        H14, Hup = self.buf(self.Rp, 512), self.buf(B, self.ih, self.iw, 512)
        self.named.update(h=Hup)
        self._linear("head.0", F, D, "head.0.weight", "head.0.bias", H14, 512, self.Rp)
        self.fwd.add("ksmi_bilinear_forward", lambda: (H14.data_ptr(), None, Hup.data_ptr(), B, gh, gw, self.ih, self.iw, 512, dt),
                     self._elt_meta("bilinear", 2 * Hup.numel()))
        self.fwd.add("ksmi_relu_forward", lambda: (Hup.data_ptr(), Hup.data_ptr(), Hup.numel(), dt), self._elt_meta("relu", 2 * Hup.numel()))
        self._conv1x1("head.2", Hup, 512, "head.2.weight", "head.2.bias", L, self.Cs, self.ncls, rows)

        def bwd():
            dHup, dH14 = self.buf(*Hup.shape), self.buf(self.Rp, 512)
            self._conv1x1_bwd("head.2", Hup, 512, "head.2.weight", "head.2.bias", dL, self.Cs, self.ncls, rows, dHup, mask=Hup)
            self.bwd.add("ksmi_bilinear_backward", lambda: (dHup.data_ptr(), dH14.data_ptr(), 0, B, gh, gw, self.ih, self.iw, 512, dt),
                         self._elt_meta("bilinear_bwd", 2 * dHup.numel()))
            self._linear_bwd("head.0", F, D, "head.0.weight", "head.0.bias", dH14, 512, self.Rp, dF if self.train_encoder else None)
        return bwd

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
