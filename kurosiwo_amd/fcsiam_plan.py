"""Static launch plan of one FC-Siam-conc / FC-Siam-diff forward/backward (row N2 of SURVEY.md §8).

Reference computation: /root/reference/models/siam_conc.py:97-177, siam_diff.py:95-173.  One "unit" = conv -> BatchNorm -> ReLU ->
Dropout2d(0.2):
  z   = conv3x3(x) + b                          implicit-GEMM kernel, BatchNorm sums in its epilogue
  out = relu(z*scale + shift) * plane mask      ksmi_bn_relu_drop2d (mask = counter-based draw of (sample, channel))
Encoder units are nn.Conv2d, decoder units nn.ConvTranspose2d(k3, p1) = the input-gradient form of a convolution (flipped taps,
reduction-major weight [Cin][Cout][3][3]): forward through the `dgrad` packing, input gradient as a plain convolution with the
weight read as [N = Cin][K = Cout], weight gradient as the convolution weight gradient with the roles of input and output
gradient swapped (one launch per concatenated source: the rows of dW follow the concat order).  The stride-2 `upconv` layers are
the four 2x2 phase convolutions of a stride-2 input gradient; their backward is a stride-2 convolution.  The shared encoder runs
once per date (its BatchNorm normalises each date on its own, siam_conc.py:100-146); weight gradients of the second pass accumulate.
"""
import ctypes as C
import os

import torch

from .changeformer_plan import CS, drop_threshold
from .fcsiam import DECODER, ENCODER
from .runtime import SrcSpec, make_conv, make_wgrad
from .snunet_plan import _Saved
from .unet_plan import UnetPlan

LAYERS = [n for st in ENCODER for n, _ in st] + [n for _, _, ch in DECODER for n, _ in ch]     # Dropout2d site = 2 * index + date


class FCSiamPlan(UnetPlan):
    input_names = ("x1", "x2")

    side_wgrad = False         # measured: many short launches, the fork events cost more than the overlap returns (4346 -> 4214 tiles/s)

    def __init__(self, model, B, H, W, dtype, training, with_backward):
        self._init_base(model, dtype, with_backward)
        self.B, self.H, self.W, self.training = B, H, W, training
        self.cin, self.nc, self.diff = model.input_nbr, model.label_nbr, model.diff
        self.x = torch.empty((2, B, self.cin, H, W), dtype=torch.float32, device=self.dev)
        self.xA, self.xB = self.x[0], self.x[1]
        self.logits = torch.empty((B, self.nc, H, W), dtype=torch.float32, device=self.dev)       # the model output (softmax / log-softmax map)
        self.dlogits = torch.empty_like(self.logits) if with_backward else None
        self.const = torch.zeros((2, 512), dtype=torch.float32, device=self.dev)
        self.const[1].fill_(1.0)
        self._gbuf, self._gacc, self._bwd, self._zeroed = {}, set(), [], set()
        p = float(model.drop2d) if training else 0.0
        self.thr, self.inv = drop_threshold(p)
        self.rng_ptr = model.rng_state().data_ptr() if self.thr else None
        if self.thr:
            self.fwd.add("ksmi_rng_advance", lambda: (self.rng_ptr,))
        self._build_fcsiam()
        if with_backward:
            for f in reversed(self._bwd):
                f()
        self._finish()

    # ---------------------------------------------------------------- building blocks
    def _bnrelu_bwd_scaled(self, bnkey, dout, out, z, sv, dz, npix, Cc):
        """out = relu(bn(z)) * Dropout2d materialised: the plane scale 1/(1-p) is constant on the active set (read from out > 0),
        so the BatchNorm + ReLU backward kernels run unchanged and alpha scales dz, dgamma, dbeta"""
        rows = max(1, min(512, npix // 256))
        self.need("bnp", rows * 2 * Cc * 4)
        self.need("bnsum", 2 * Cc * 4)
        gw, gb = self.m._g(f"{bnkey}.weight").data_ptr(), self.m._g(f"{bnkey}.bias").data_ptr()
        a1, _ = self._acc_param(f"{bnkey}.weight"), self._acc_param(f"{bnkey}.bias")
        gamma = self.m._p(f"{bnkey}.weight").data_ptr()
        dt, alpha = self.dt, C.c_float(self.inv)
        self.bwd.add("ksmi_bnrelu_bwd_reduce", lambda: (dout.data_ptr(), out.data_ptr(), z.data_ptr(), sv.mean, sv.rstd, self.scr("bnp"), rows, npix, Cc, dt),
                     self._elt_meta("bnrelu_bwd_reduce", 3 * npix * Cc))
        self.bwd.add("ksmi_reduce_rows_scaled", lambda: (self.scr("bnp"), rows, 2, Cc, Cc, self.scr("bnsum"), gw, gb, a1, alpha))
        self._mark(f"{bnkey}.weight", f"{bnkey}.bias")
        self.bwd.add("ksmi_bnrelu_bwd_apply_scaled", lambda: (dout.data_ptr(), out.data_ptr(), z.data_ptr(), sv.mean, sv.rstd, gamma, self.scr("bnsum"),
                                                              dz.data_ptr(), float(npix), npix, Cc, alpha, dt),
                     self._elt_meta("bnrelu_bwd_apply", 5 * npix * Cc))

    def _zero_bias(self, bkey):
        """a convolution bias followed directly by a train-mode BatchNorm has an analytically zero gradient"""
        if bkey not in self._zeroed:
            self._zeroed.add(bkey)
            self._zero_grad_key(bkey)

    def _unit(self, name, srcs, Ktot, Cout, h, w, date, transposed, image=False):
        """srcs: [(tensor, channels)] in concat order (image=True: the NHWC copy of the input tile with zero pad channels)"""
        m, B, dt = self.m, self.B, self.dt
        npix = B * h * w
        z, out = self.buf(B, h, w, Cout), self.buf(B, h, w, Cout)
        sv = _Saved(max(Cout, 16), self.dev)
        wkey, bkey, bnkey = f"conv{name}.weight", f"conv{name}.bias", f"bn{name}"
        specs = [SrcSpec(t, t.shape[-1], k_real=self.cin) for t, c in srcs] if image else [SrcSpec(t, c) for t, c in srcs]
        if transposed:
            rows, cpad = self._conv3(self.fwd, f"conv{name}", specs, [(z, Cout, 0, 0, Cout, 0)], wkey, bkey, B, h, w, Cout, Ktot,
                                     stats=self.training, dgrad=True)
        else:
            rows, cpad = self._cv(self.fwd, f"conv{name}", specs, [(z, Cout, 0, 0, Cout, 0)], wkey, h, w, h, w, 3, 1, 1, Cout, Ktot,
                                  stats=self.training, bias=m._p(bkey))
        self._bn_finalize(bnkey, sv, rows, cpad, Cout, npix)
        site = 2 * LAYERS.index(name) + date
        self.fwd.add("ksmi_bn_relu_drop2d", lambda: (z.data_ptr(), sv.scale, sv.shift, out.data_ptr(), B, h * w, Cout, self.thr, C.c_float(self.inv), site,
                                                     self.rng_ptr, dt), self._elt_meta("bn_relu_drop2d", 2 * npix * Cout))
        self.named[f"{name}_{date + 1}" if not transposed else name] = out

        def bwd():
            dout = self.gbuf(out)
            dz = self.buf(B, h, w, Cout)
            self._bnrelu_bwd_scaled(bnkey, dout, out, z, sv, dz, npix, Cout)
            self._zero_bias(bkey)
            if transposed:
                acc, off = self._acc_param(wkey), 0
                for t, c in srcs:                       # dWt[ci][co][k]: the convolution weight gradient with input = dz, output gradient = source
                    gview = m._g(wkey)[off * Cout * 9:(off + c) * Cout * 9]
                    self.keep.append(gview)
                    dw, ws = make_wgrad([SrcSpec(dz, Cout)], t, c, 0, c, gview, 9, Cout * 9, 1, acc, B, h, w, h, w, 3, 3, 1, 1, self.dtype)
                    self._wgrad(dw, ws, wkey)
                    off += c
                dsts, off = [], 0
                for t, c in srcs:
                    dsts.append((self.gbuf(t), c, 0, off, c, self.gacc(t)))
                    off += c
                self._cv(self.bwd, f"conv{name}.dgrad", [SrcSpec(dz, Cout)], dsts, wkey, h, w, h, w, 3, 1, 1, Ktot, Cout, tag="dgrad")
            else:
                self._wg(specs, dz, Cout, wkey, h, w, h, w, 3, 1, 1, Ktot)
                if not image:
                    (t, c), = srcs
                    self._conv3(self.bwd, f"conv{name}", [SrcSpec(dz, Cout)], [(self.gbuf(t), c, 0, 0, c, self.gacc(t))], wkey, None, B, h, w, c, Cout,
                                dgrad=True)
        self._bwd.append(bwd)
        return out

    def _pool2(self, x, Cc, h, w):
        B, dt = self.B, self.dt
        y = self.buf(B, h // 2, w // 2, Cc)
        self.fwd.add("ksmi_maxpool2x2_forward", lambda: (x.data_ptr(), y.data_ptr(), B, h, w, Cc, dt), self._elt_meta("maxpool2", 5 * B * h * w * Cc // 4))

        def bwd():
            dy, dx = self.gbuf(y), self.gbuf(x)
            acc = self.gacc(x)
            self.bwd.add("ksmi_maxpool2x2_backward", lambda: (x.data_ptr(), dy.data_ptr(), dx.data_ptr(), acc, B, h, w, Cc, dt),
                         self._elt_meta("maxpool2_bwd", 9 * B * h * w * Cc // 4))
        self._bwd.append(bwd)
        return y

    def _upconv(self, lvl, x, Cu, h, w):
        """nn.ConvTranspose2d(Cu, Cu, 3, stride=2, padding=1, output_padding=1): y[2m+py][2n+px] as four 2x2 phase convolutions of x"""
        m, B = self.m, self.B
        wkey, bkey = f"upconv{lvl}.weight", f"upconv{lvl}.bias"
        y = self.buf(B, 2 * h, 2 * w, Cu)
        for py in range(2):
            for px in range(2):
                tap_map = []
                for a in range(2):
                    for b in range(2):
                        ky = (1 if a == 0 else -1) if py == 0 else (2 if a == 0 else 0)
                        kx = (1 if b == 0 else -1) if px == 0 else (2 if b == 0 else 0)
                        tap_map.append(-1 if ky < 0 or kx < 0 else ky * 3 + kx)
                d, table = make_conv([SrcSpec(x, Cu)], [(y, Cu, 0, 0, Cu, 0)], y, m._p(bkey), None, B, h, w, h, w, 2, 2, 1, 0, Cu, self.dtype,
                                     out_map=(2, 2, py, px, 2 * h, 2 * w))
                d.wpk = self._packed(wkey, table, 4, Cu, Cu, Cu * 9, 9, 0, 1, 0, tap_map).data_ptr()
                self._conv(self.fwd, d, "upconv_phase", f"upconv{lvl}.p{py}{px}")

        def bwd():
            dy = self.gbuf(y)
            self._wg([SrcSpec(dy, Cu)], x, Cu, wkey, 2 * h, 2 * w, h, w, 3, 2, 1, Cu)
            self._cv(self.bwd, f"upconv{lvl}.dgrad", [SrcSpec(dy, Cu)], [(self.gbuf(x), Cu, 0, 0, Cu, self.gacc(x))], wkey, 2 * h, 2 * w, h, w, 3, 2, 1, Cu, Cu,
                     tag="dgrad")
            self._bias_grad(dy, B * 4 * h * w, Cu, bkey)
        self._bwd.append(bwd)
        return y

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
    def _build_fcsiam(self):
        m, B, H, W, dt, nc = self.m, self.B, self.H, self.W, self.dt, self.nc
        # channel stride of the NHWC image copy: one 16-byte granule is enough since the persistent short-K kernel reads partial chunks
        # (igemm3.hip klen, round 5: 2 real channels of 8 instead of 32 -- the copy and conv11 move a quarter of the bytes);
        # KSMI_IGEMM3_PARTIAL=0 (the A/B switch of that rule) restores whole chunks
        kc = 32 if self.dtype == torch.bfloat16 else 16
        if self.dtype == torch.bfloat16 and os.environ.get("KSMI_IGEMM3_PARTIAL", "1") != "0":
            kc = 8
        Kpad = -(-self.cin // kc) * kc
        skips = []
        deepest = None
        for date in range(2):
            img = self.buf(B, H, W, Kpad)                         # NHWC copy of the fp32 NCHW tile, zero pad channels
            self.fwd.add("ksmi_im2col", lambda date=date, img=img: (self.x[date].data_ptr(), img.data_ptr(), B, self.cin, H, W, H, W, 1, 1, 1, 0, Kpad, 1, dt),
                         self._elt_meta("to_nhwc", 2 * B * H * W * Kpad))
            t, c, h, w = img, self.cin, H, W
            sk = []
            for si, stage in enumerate(ENCODER):
                for li, (name, co) in enumerate(stage):
                    t = self._unit(name, [(t, c)], c, co, h, w, date, False, image=(si == 0 and li == 0))
                    c = co
                sk.append((t, c, h, w))
                if date == 1 or si < len(ENCODER) - 1:            # the pooled stage-4 map of date 1 has no consumer (siam_conc.py:148-150)
                    t = self._pool2(t, c, h, w)
                    h, w = h // 2, w // 2
            skips.append(sk)
            deepest = (t, c, h, w)
        y, cy, h, w = deepest
        for lvl, cu, chain in DECODER:
            up = self._upconv(lvl, y, cu, h, w)
            h, w = 2 * h, 2 * w
            (s1, c1, _, _), (s2, _, _, _) = skips[0][lvl - 1], skips[1][lvl - 1]
            srcs = [(up, cu), (self._absdiff(s1, s2, c1, h, w), c1)] if self.diff else [(up, cu), (s1, c1), (s2, c1)]
            y, cy = None, sum(c for _, c in srcs)
            for name, co in chain:
                y = self._unit(name, srcs, cy, co, h, w, 0, True)
                srcs, cy = [(y, co)], co
        # conv11d (ConvTranspose2d 16 -> label_nbr) + Softmax / LogSoftmax over the classes
        P = self.buf(B, H, W, CS)
        wk, bk = "conv11d.weight", "conv11d.bias"
        self._conv3(self.fwd, "conv11d", [SrcSpec(y, 16)], [(P, CS, 0, 0, nc, 0)], wk, bk, B, H, W, nc, 16, dgrad=True)
        HW, act = H * W, (3 if self.diff else 2)
        self.fwd.add("ksmi_out_to_nchw", lambda: (P.data_ptr(), self.logits.data_ptr(), B, nc, CS, HW, act, dt))
        x12d = y

        def head_bwd():
            dP = self.buf(B * HW, CS)
            self.bwd.add("ksmi_dout_to_nhwc", lambda: (self.dlogits.data_ptr(), self.logits.data_ptr(), dP.data_ptr(), B, nc, CS, HW, act, dt))
            psrc = [SrcSpec(dP, CS, 0, CS, k_real=nc)]
            self._wg(psrc, x12d, 16, wk, H, W, H, W, 3, 1, 1, nc)
            self._cv(self.bwd, "conv11d.dgrad", psrc, [(self.gbuf(x12d), 16, 0, 0, 16, self.gacc(x12d))], wk, H, W, H, W, 3, 1, 1, 16, nc, tag="dgrad")
            rr = max(1, min(512, B * HW // 256))
            self.need("red", rr * CS * 4)
            accb = self._acc_param(bk)
            gb = m._g(bk).data_ptr()
            self.bwd.add("ksmi_channel_sum", lambda: (dP.data_ptr(), self.scr("red"), rr, B * HW, CS, dt), self._elt_meta("channel_sum", B * HW * CS))
            self.bwd.add("ksmi_reduce_rows", lambda: (self.scr("red"), rr, 1, CS, nc, None, None, gb, accb))
            self._mark(bk)
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
