"""Static launch plan of one BIT-CD (`base_resnet18`) forward/backward (row N2 of SURVEY.md §8).

Reference computation: /root/reference/models/bit_cd.py `ResNet.forward` :763-778 / `forward_single` :780-797.  The siamese ResNet-18
runs once per date through the BasicBlock builder of the U1 row (kurosiwo_amd/unet_plan.py: conv -> BN -> ReLU chains fused into the
consumers' operand loads and input-gradient epilogues); weight gradients of the second pass accumulate, BatchNorm statistics are
per date.  Head: nearest x2, conv_pred, |f1 - f2|, bilinear x4, conv3x3 -> BN -> ReLU -> conv3x3.
"""
import torch

from .bitcd import LAYERS
from .changeformer_plan import CS
from .runtime import SrcSpec, conv_grid_m, make_conv, make_wgrad
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
        rows0 = conv_grid_m(d)
        if self.training:
            self.need("stats", rows0 * 2 * d.Npad * 4)
            self._later.append(lambda: setattr(d, "stats", self.scr("stats")))
        self._conv(self.fwd, d, "stem7x7", "resnet.conv1")
        self._bn_finalize("resnet.bn1", sv0, rows0, d.Npad, 64, R1)
        self._affine(self.fwd, s0, sv0, f1, R1, 64, 1)
        H2, W2 = H1 // 2, W1 // 2
        p = self.buf(B, H2, W2, 64)
        self.fwd.add("ksmi_maxpool3x3s2_forward", lambda: (f1.data_ptr(), p.data_ptr(), B, H1, W1, 64, dt), self._elt_meta("maxpool3", 2 * R1 * 64))

        def bwd():
            df1, ds0 = self.gbuf(f1), self.buf(R1, 64)
            dp = self.gbuf(p)
            acc = self.gacc(f1)
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
    def _build_bitcd(self):
        m, B, H, W, dt, nc = self.m, self.B, self.H, self.W, self.dt, self.nc
        preds = []
        for date in range(2):
            t, h, w = self._stem(date)
            cin = 64
            for li, (ch, stride) in enumerate(LAYERS):
                for bi in range(2):
                    s_ = stride if bi == 0 else 1
                    t = self._basic_block(f"resnet.layer{li + 1}.{bi}", t, cin, ch, h, w, s_)
                    h, w, cin = h // s_, w // s_, ch
                self.named[f"layer{li + 1}_{date + 1}"] = t
            # nn.Upsample(scale_factor=2) (nearest) -> conv_pred
            U = self.buf(B, 2 * h, 2 * w, 512)
            self.fwd.add("ksmi_upsample2_forward", lambda t=t, U=U, h=h, w=w: (t.data_ptr(), U.data_ptr(), B, h, w, 512, 0, dt),
                         self._elt_meta("upsample2", 5 * B * h * w * 512))
            h2, w2 = 2 * h, 2 * w
            Pd = self.buf(B, h2, w2, 32)
            self._cv(self.fwd, "conv_pred", [SrcSpec(U, 512)], [(Pd, 32, 0, 0, 32, 0)], "conv_pred.weight", h2, w2, h2, w2, 3, 1, 1, 32, 512,
                     bias=m._p("conv_pred.bias"))
            self.named[f"pred_{date + 1}"] = Pd

            def bwd(t=t, U=U, Pd=Pd, h=h, w=w, h2=h2, w2=w2):
                dPd, dU = self.gbuf(Pd), self.buf(B, h2, w2, 512)
                self._wg([SrcSpec(U, 512)], dPd, 32, "conv_pred.weight", h2, w2, h2, w2, 3, 1, 1, 512)
                self._conv3(self.bwd, "conv_pred", [SrcSpec(dPd, 32)], [(dU, 512, 0, 0, 512, 0)], "conv_pred.weight", None, B, h2, w2, 512, 32, dgrad=True)
                self._bias_grad(dPd, B * h2 * w2, 32, "conv_pred.bias")
                dt_ = self.gbuf(t)
                if self.gacc(t):
                    raise RuntimeError("unexpected second writer of a backbone output gradient")
                self.bwd.add("ksmi_upsample2_backward", lambda: (dU.data_ptr(), None, dt_.data_ptr(), B, h, w, 512, 0, dt),
                             self._elt_meta("upsample2_bwd", 5 * B * h * w * 512))
            self._bwd.append(bwd)
            preds.append((Pd, h2, w2))
        (f1, h2, w2), (f2, _, _) = preds
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
            for key in ("resnet.fc.weight", "resnet.fc.bias"):       # the unused ImageNet head of the backbone: no gradient
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
