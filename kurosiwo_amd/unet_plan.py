"""Static launch plan of one Unet(resnet18) forward/backward (row U1 of SURVEY.md §8(a)); architecture: oracle/unet_ref.py.

conv -> BatchNorm -> ReLU chains never materialise the normalised tensor when the consumer is a convolution (BN-apply + ReLU on the
operand load; the consumer's input-gradient epilogue applies the ReLU mask and accumulates the BN-backward sums); tensors that feed
pooling / upsampling / skip concatenation / the residual sum are materialised once.  Strided 3x3 and 1x1 convolutions get their
input gradients as phase convolutions written through the strided output placement of the implicit-GEMM kernel.
"""
import ctypes as C

import os

import torch

from . import _lib
from .changeformer_plan import BN_EPS, BN_MOMENTUM, CS, ChangeFormerPlan
from .runtime import SrcSpec, conv_grid_m, conv_stats_rows, make_conv, make_wgrad
from .snunet_plan import _Saved
from .unet import DECODER_CHANNELS, LAYERS


class UnetPlan(ChangeFormerPlan):
    slab_bias_side = False     # (measured on ChangeFormer only: plan_base._linear_wgrad)
    input_names = ("x",)

    side_tokens = False        # (ChangeFormerPlan's encoder switch: no MiT encoder here)
    side_wgrad = True          # plan_base.PlanBase.side_wgrad: dedicated buffers throughout (self.buf), conv weight gradients only

    def __init__(self, model, B, H, W, dtype, training, with_backward):
        self._init_base(model, dtype, with_backward)
        self.B, self.H, self.W, self.training = B, H, W, training
        self.cin, self.nc = model.in_channels, model.classes
        self.x = torch.empty((B, self.cin, H, W), dtype=torch.float32, device=self.dev)
        self.logits = torch.empty((B, self.nc, H, W), dtype=torch.float32, device=self.dev)
        self.dlogits = torch.empty_like(self.logits) if with_backward else None
        self.const = torch.zeros((2, 512), dtype=torch.float32, device=self.dev)
        self.const[1].fill_(1.0)
        self._gbuf = {}            # id(tensor) -> gradient buffer ; first writer "=", later writers "+="
        self._gacc = set()
        self._bwd = []
        self._build_unet()
        if with_backward:
            for f in reversed(self._bwd):
                f()
        self._finish()

    # ---------------------------------------------------------------- gradient bookkeeping
    def gbuf(self, t):
        if id(t) not in self._gbuf:
            self._gbuf[id(t)] = self.buf(*t.shape)
        return self._gbuf[id(t)]

    def gacc(self, t):
        """accumulate flag for the next writer of d(t): 0 for the first one"""
        a = 1 if id(t) in self._gacc else 0
        self._gacc.add(id(t))
        return a

    # ---------------------------------------------------------------- building blocks
    def _cv(self, ll, name, srcs, dsts, wkey, Hin, Win, Hout, Wout, k, stride, pad, N, Ktot, stats=False, mask=None, bias=None, tag="conv"):
        d, table = make_conv(srcs, dsts, dsts[0][0], bias, None, self.B, Hin, Win, Hout, Wout, k, k, stride, pad, N, self.dtype, mask=mask)
        taps = k * k
        d.wpk = self._packed(wkey, table, taps, N, N, taps, Ktot * taps, 0, 1, 0).data_ptr()
        rows = conv_stats_rows(d, self.dtype) if stats else conv_grid_m(d)   # (rows of the kernel that will run it: see changeformer_plan._conv3)
        if stats:
            self.need("stats", rows * 2 * d.Npad * 4)
            self._later.append(lambda: setattr(d, "stats", self.scr("stats")))
        self._conv(ll, d, f"{tag}{k}x{k}", name)
        return rows, d.Npad

    def _wg(self, srcs, dy, N, wkey, Hin, Win, Hout, Wout, k, stride, pad, Ktot):
        taps = k * k
        dw, ws = make_wgrad(srcs, dy, N, 0, N, self.m._g(wkey), taps, Ktot * taps, 1, self._acc_param(wkey), self.B, Hin, Win, Hout, Wout,
                            k, k, stride, pad, self.dtype)
        self._wgrad(dw, ws, wkey)

    def _dgrad_s2(self, name, dy, Cout, dx, Cin, H, W, wkey, acc):
        """input gradient of a 3x3 stride-2 pad-1 convolution: dx[2m] = dy[m] W[1] ; dx[2m+1] = dy[m] W[2] + dy[m+1] W[0] per axis
        -> four 2x2 phase convolutions over dy (a missing tap packs as zero) placed at (2m+py, 2n+px)"""
        Ho, Wo = H // 2, W // 2
        for py in range(2):
            for px in range(2):
                tap_map = []
                for a in range(2):
                    for b in range(2):
                        ky = (1 if a == 0 else -1) if py == 0 else (2 if a == 0 else 0)
                        kx = (1 if b == 0 else -1) if px == 0 else (2 if b == 0 else 0)
                        tap_map.append(-1 if ky < 0 or kx < 0 else ky * 3 + kx)
                d, table = make_conv([SrcSpec(dy, Cout)], [(dx, Cin, 0, 0, Cin, acc)], dx, None, None, self.B, Ho, Wo, Ho, Wo, 2, 2, 1, 0, Cin,
                                     self.dtype, out_map=(2, 2, py, px, H, W))
                d.wpk = self._packed(wkey, table, 4, Cin, Cin, Cin * 9, 9, 0, 1, 0, tap_map).data_ptr()
                self._conv(self.bwd, d, "dgrad_s2_phase", f"{name}.p{py}{px}")

    def _bnrelu_bwd(self, bnkey, dout, out, z, sv, dz, npix, Cc):
        """out = relu(bn(z) [+ identity]) materialised: dout -> g (in place), dz, dgamma, dbeta"""
        rows = max(1, min(512, npix // 256))
        self.need("bnp", rows * 2 * Cc * 4)
        self.need("bnsum", 2 * Cc * 4)
        gw, gb = self.m._g(f"{bnkey}.weight").data_ptr(), self.m._g(f"{bnkey}.bias").data_ptr()
        a1, _ = self._acc_param(f"{bnkey}.weight"), self._acc_param(f"{bnkey}.bias")
        gamma = self.m._p(f"{bnkey}.weight").data_ptr()
        dt = self.dt
        self.bwd.add("ksmi_bnrelu_bwd_reduce", lambda: (dout.data_ptr(), out.data_ptr(), z.data_ptr(), sv.mean, sv.rstd, self.scr("bnp"), rows, npix, Cc, dt),
                     self._elt_meta("bnrelu_bwd_reduce", 3 * npix * Cc))
        # the partial rows finish inside the apply pass (bnfused.hip, as in the SNUNet plan since round 4): one launch less per BatchNorm;
        # KSMI_BN_FUSED_FAMILIES=0 keeps the separate reduce_rows launch (A/B)
        if os.environ.get("KSMI_BN_FUSED_FAMILIES", "1") != "0" and self.lib.ksmi_bn_fused_supported(Cc, Cc, dt):
            self.bwd.add("ksmi_bnrelu_bwd_fin_apply", lambda: (self.scr("bnp"), rows, Cc, self.scr("bnsum"), gw, gb, a1, dout.data_ptr(), out.data_ptr(),
                                                               z.data_ptr(), sv.mean, sv.rstd, gamma, dz.data_ptr(), float(npix), npix, Cc, dt),
                         self._elt_meta("bnrelu_bwd_apply", 5 * npix * Cc))
            self._mark(f"{bnkey}.weight", f"{bnkey}.bias")
            return
        self.bwd.add("ksmi_reduce_rows", lambda: (self.scr("bnp"), rows, 2, Cc, Cc, self.scr("bnsum"), gw, gb, a1))
        self._mark(f"{bnkey}.weight", f"{bnkey}.bias")
        self.bwd.add("ksmi_bnrelu_bwd_apply", lambda: (dout.data_ptr(), out.data_ptr(), z.data_ptr(), sv.mean, sv.rstd, gamma, self.scr("bnsum"), dz.data_ptr(),
                                                       float(npix), npix, Cc, dt), self._elt_meta("bnrelu_bwd_apply", 5 * npix * Cc))

    def _bn_plain_bwd(self, bnkey, dy, x, sv, dv, npix, Cc):
        rows = max(1, min(512, npix // 256))
        self.need("bnp", rows * 2 * Cc * 4)
        self.need("bnsum", 2 * Cc * 4)
        gw, gb = self.m._g(f"{bnkey}.weight").data_ptr(), self.m._g(f"{bnkey}.bias").data_ptr()
        a1, _ = self._acc_param(f"{bnkey}.weight"), self._acc_param(f"{bnkey}.bias")
        gamma = self.m._p(f"{bnkey}.weight").data_ptr()
        dt = self.dt
        self.bwd.add("ksmi_bn_bwd_reduce", lambda: (dy.data_ptr(), x.data_ptr(), sv.mean, sv.rstd, self.scr("bnp"), rows, npix, Cc, dt),
                     self._elt_meta("bn_bwd_reduce", 2 * npix * Cc))
        self.bwd.add("ksmi_reduce_rows", lambda: (self.scr("bnp"), rows, 2, Cc, Cc, self.scr("bnsum"), gw, gb, a1))
        self._mark(f"{bnkey}.weight", f"{bnkey}.bias")
        self.bwd.add("ksmi_bn_bwd_apply", lambda: (dy.data_ptr(), x.data_ptr(), sv.mean, sv.rstd, gamma, self.scr("bnsum"), dv.data_ptr(), 0, float(npix),
                                                   npix, Cc, dt), self._elt_meta("bn_bwd_apply", 3 * npix * Cc))

    def _bnmask(self, z, sv):
        """epilogue mask of a consumer's input gradient: ReLU(bn(z)) active set + BatchNorm-backward sums"""
        return (z, sv.t[0], sv.t[1], sv.t[2], sv.t[3])

    def _affine(self, ll, x, sv, y, npix, Cc, relu):
        dt = self.dt
        ll.add("ksmi_affine", lambda: (x.data_ptr(), sv.scale, sv.shift, y.data_ptr(), npix, Cc, relu, C.c_float(1.0), dt),
               self._elt_meta("bn_apply", 2 * npix * Cc))

    # ---------------------------------------------------------------- BasicBlock (torchvision resnet.py)
    def _basic_block(self, k, x_in, Cin, Cout, H, W, stride):
        B, dt = self.B, self.dt
        Ho, Wo = H // stride, W // stride
        npix = B * Ho * Wo
        i1, z2, out = self.buf(B, Ho, Wo, Cout), self.buf(B, Ho, Wo, Cout), self.buf(B, Ho, Wo, Cout)
        sv1, sv2 = _Saved(Cout, self.dev), _Saved(Cout, self.dev)
        rows, cpad = self._cv(self.fwd, f"{k}.conv1", [SrcSpec(x_in, Cin)], [(i1, Cout, 0, 0, Cout, 0)], f"{k}.conv1.weight", H, W, Ho, Wo, 3, stride, 1,
                              Cout, Cin, stats=self.training)
        self._bn_finalize(f"{k}.bn1", sv1, rows, cpad, Cout, npix)
        a1 = [SrcSpec(i1, Cout, scale=sv1.scale_t, shift=sv1.shift_t, relu=1)]
        rows, cpad = self._cv(self.fwd, f"{k}.conv2", a1, [(z2, Cout, 0, 0, Cout, 0)], f"{k}.conv2.weight", Ho, Wo, Ho, Wo, 3, 1, 1, Cout, Cout,
                              stats=self.training)
        rows2, cpad2 = rows, cpad
        # bn2: its statistics finish inside the apply pass below when that is available (bnfused.hip: one launch instead of two; the
        # downsample branch in between writes other rows of the shared statistics scratch, so it needs its own finalize first)
        down = f"{k}.downsample.0.weight" in self.m._pspec
        fuse2 = (self.training and not down and os.environ.get("KSMI_BN_FUSED_FAMILIES", "1") != "0"
                 and bool(self.lib.ksmi_bn_fused_supported(Cout, cpad2, dt)))
        if not fuse2:
            self._bn_finalize(f"{k}.bn2", sv2, rows, cpad, Cout, npix)
        if down:
            ds, idn = self.buf(B, Ho, Wo, Cout), self.buf(B, Ho, Wo, Cout)
            svd = _Saved(Cout, self.dev)
            rows, cpad = self._cv(self.fwd, f"{k}.downsample", [SrcSpec(x_in, Cin)], [(ds, Cout, 0, 0, Cout, 0)], f"{k}.downsample.0.weight", H, W, Ho, Wo,
                                  1, stride, 0, Cout, Cin, stats=self.training)
            self._bn_finalize(f"{k}.downsample.1", svd, rows, cpad, Cout, npix)
            self._affine(self.fwd, ds, svd, idn, npix, Cout, 0)
        else:
            idn = x_in
        if fuse2:
            m = self.m
            kb = f"{k}.bn2"
            g2, b2 = m._p(f"{kb}.weight").data_ptr(), m._p(f"{kb}.bias").data_ptr()
            rm, rv, nbt = m._b(f"{kb}.running_mean").data_ptr(), m._b(f"{kb}.running_var").data_ptr(), m._c(f"{kb}.num_batches_tracked").data_ptr()
            st = self._stats_ptr()
            self.fwd.add("ksmi_bn_fin_add_relu", lambda: (st(), rows2, cpad2, Cout, float(npix), g2, b2, rm, rv, nbt, BN_MOMENTUM, BN_EPS,
                                                          sv2.mean, sv2.rstd, sv2.scale, sv2.shift, z2.data_ptr(), idn.data_ptr(), out.data_ptr(),
                                                          None, B, Ho, Wo, dt), self._elt_meta("bn_add_relu", 3 * npix * Cout))
        else:
            self.fwd.add("ksmi_bn_add_relu", lambda: (z2.data_ptr(), idn.data_ptr(), sv2.scale, sv2.shift, out.data_ptr(), npix, Cout, dt),
                         self._elt_meta("bn_add_relu", 3 * npix * Cout))

        def bwd():
            dout = self.gbuf(out)
            dz2, da1, di1 = self.buf(B, Ho, Wo, Cout), self.buf(B, Ho, Wo, Cout), self.buf(B, Ho, Wo, Cout)
            self._bnrelu_bwd(f"{k}.bn2", dout, out, z2, sv2, dz2, npix, Cout)            # dout now holds g = dout * (out > 0)
            # conv2
            self._wg(a1, dz2, Cout, f"{k}.conv2.weight", Ho, Wo, Ho, Wo, 3, 1, 1, Cout)
            r2, c2 = self._conv3(self.bwd, f"{k}.conv2", [SrcSpec(dz2, Cout)], [(da1, Cout, 0, 0, Cout, 0)], f"{k}.conv2.weight", None, B, Ho, Wo, Cout, Cout,
                                 mask=self._bnmask(i1, sv1), stats=True, dgrad=True)
            self._bn_backward(f"{k}.bn1", da1, i1, sv1, di1, r2, c2, Cout, npix, npix, 0)
            # conv1
            self._wg([SrcSpec(x_in, Cin)], di1, Cout, f"{k}.conv1.weight", H, W, Ho, Wo, 3, stride, 1, Cin)
            dx = self.gbuf(x_in)
            acc = self.gacc(x_in)
            if stride == 1:
                self._conv3(self.bwd, f"{k}.conv1", [SrcSpec(di1, Cout)], [(dx, Cin, 0, 0, Cin, acc)], f"{k}.conv1.weight", None, B, H, W, Cin, Cout, dgrad=True)
            else:
                self._dgrad_s2(f"{k}.conv1", di1, Cout, dx, Cin, H, W, f"{k}.conv1.weight", acc)
            # identity branch
            if down:
                dds = self.buf(B, Ho, Wo, Cout)
                self._bn_plain_bwd(f"{k}.downsample.1", dout, ds, svd, dds, npix, Cout)
                self._wg([SrcSpec(x_in, Cin)], dds, Cout, f"{k}.downsample.0.weight", H, W, Ho, Wo, 1, stride, 0, Cin)
                d, table = make_conv([SrcSpec(dds, Cout)], [(dx, Cin, 0, 0, Cin, 1)], dx, None, None, B, Ho, Wo, Ho, Wo, 1, 1, 1, 0, Cin, self.dtype,
                                     out_map=(stride, stride, 0, 0, H, W) if stride != 1 else None)   # (stride 1: BIT-CD's layer3 / layer4)
                d.wpk = self._packed(f"{k}.downsample.0.weight", table, 1, Cin, Cin, Cin, 1, 0, 0).data_ptr()
                self._conv(self.bwd, d, "dgrad_1x1s2", f"{k}.downsample")
            else:
                self.bwd.add("ksmi_add", lambda: (dx.data_ptr(), dout.data_ptr(), dx.data_ptr(), npix * Cout, dt), self._elt_meta("add", 3 * npix * Cout))
        self._bwd.append(bwd)
        return out

    # ---------------------------------------------------------------- DecoderBlock (smp unet/decoder.py)
    def _decoder_block(self, k, xd, Cin, skip, Cs, Cout, h, w):
        B, dt = self.B, self.dt
        H2, W2 = 2 * h, 2 * w
        npix = B * H2 * W2
        U = self.buf(B, H2, W2, Cin)
        self.fwd.add("ksmi_upsample2_forward", lambda: (xd.data_ptr(), U.data_ptr(), B, h, w, Cin, 0, dt), self._elt_meta("upsample2", 5 * B * h * w * Cin))
        z1, z2, y = self.buf(B, H2, W2, Cout), self.buf(B, H2, W2, Cout), self.buf(B, H2, W2, Cout)
        svA, svB = _Saved(max(Cout, 16), self.dev), _Saved(max(Cout, 16), self.dev)
        srcs = [SrcSpec(U, Cin)] + ([SrcSpec(skip, Cs)] if skip is not None else [])
        Kt = Cin + (Cs if skip is not None else 0)
        rows, cpad = self._cv(self.fwd, f"{k}.conv1", srcs, [(z1, Cout, 0, 0, Cout, 0)], f"{k}.conv1.0.weight", H2, W2, H2, W2, 3, 1, 1, Cout, Kt, stats=self.training)
        self._bn_finalize(f"{k}.conv1.1", svA, rows, cpad, Cout, npix)
        a1 = [SrcSpec(z1, Cout, scale=svA.scale_t, shift=svA.shift_t, relu=1)]
        rows, cpad = self._cv(self.fwd, f"{k}.conv2", a1, [(z2, Cout, 0, 0, Cout, 0)], f"{k}.conv2.0.weight", H2, W2, H2, W2, 3, 1, 1, Cout, Cout, stats=self.training)
        self._bn_finalize(f"{k}.conv2.1", svB, rows, cpad, Cout, npix)
        self._affine(self.fwd, z2, svB, y, npix, Cout, 1)

        def bwd():
            dy = self.gbuf(y)
            dz2, da1, dz1, dU = self.buf(B, H2, W2, Cout), self.buf(B, H2, W2, Cout), self.buf(B, H2, W2, Cout), self.buf(B, H2, W2, Cin)
            self._bnrelu_bwd(f"{k}.conv2.1", dy, y, z2, svB, dz2, npix, Cout)
            self._wg(a1, dz2, Cout, f"{k}.conv2.0.weight", H2, W2, H2, W2, 3, 1, 1, Cout)
            r2, c2 = self._conv3(self.bwd, f"{k}.conv2", [SrcSpec(dz2, Cout)], [(da1, Cout, 0, 0, Cout, 0)], f"{k}.conv2.0.weight", None, B, H2, W2, Cout, Cout,
                                 mask=self._bnmask(z1, svA), stats=True, dgrad=True)
            self._bn_backward(f"{k}.conv1.1", da1, z1, svA, dz1, r2, c2, Cout, npix, npix, 0)
            self._wg(srcs, dz1, Cout, f"{k}.conv1.0.weight", H2, W2, H2, W2, 3, 1, 1, Kt)
            dsts = [(dU, Cin, 0, 0, Cin, 0)]
            if skip is not None:
                dsts.append((self.gbuf(skip), Cs, 0, Cin, Cs, self.gacc(skip)))
            self._conv3(self.bwd, f"{k}.conv1", [SrcSpec(dz1, Cout)], dsts, f"{k}.conv1.0.weight", None, B, H2, W2, Kt, Cout, dgrad=True)
            dxd = self.gbuf(xd)
            if self.gacc(xd):
                raise _lib.KsmiError("unexpected second writer of a decoder input gradient")
            self.bwd.add("ksmi_upsample2_backward", lambda: (dU.data_ptr(), None, dxd.data_ptr(), B, h, w, Cin, 0, dt), self._elt_meta("upsample2_bwd", 5 * B * h * w * Cin))
        self._bwd.append(bwd)
        return y

    # ---------------------------------------------------------------- the graph
    def _build_unet(self):
        m, B, H, W, dt = self.m, self.B, self.H, self.W, self.dt
        H1, W1 = H // 2, W // 2
        R1 = B * H1 * W1
        kc = 32 if self.dtype == torch.bfloat16 else 16
        Kreal = self.cin * 49
        Kpad = -(-Kreal // kc) * kc
        col, s0, f1 = self.buf(R1, Kpad), self.buf(R1, 64), self.buf(B, H1, W1, 64)
        sv0 = _Saved(64, self.dev)
        self.fwd.add("ksmi_im2col", lambda: (self.x.data_ptr(), col.data_ptr(), B, self.cin, H, W, H1, W1, 7, 7, 2, 3, Kpad, 1, dt), self._elt_meta("im2col", 2 * R1 * Kpad))
        d, table = make_conv([SrcSpec(col, Kpad, k_real=Kreal)], [(s0, 64, 0, 0, 64, 0)], s0, None, None, 1, R1, 1, R1, 1, 1, 1, 1, 0, 64, self.dtype)
        d.wpk = self._packed("encoder.conv1.weight", table, 1, 64, 64, 1, Kreal, 0, 0).data_ptr()
        rows0 = conv_stats_rows(d, self.dtype) if self.training else conv_grid_m(d)   # (rows of the kernel that will run it: see changeformer_plan._conv3)
        if self.training:
            self.need("stats", rows0 * 2 * d.Npad * 4)
            self._later.append(lambda: setattr(d, "stats", self.scr("stats")))
        self._conv(self.fwd, d, "stem7x7", "encoder.conv1")
        self._bn_finalize("encoder.bn1", sv0, rows0, d.Npad, 64, R1)
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

        def stem_bwd():
            df1, ds0 = self.gbuf(f1), self.buf(R1, 64)
            dp = self.gbuf(p)
            acc = self.gacc(f1)
            if pidx is not None:
                self.bwd.add("ksmi_maxpool3x3s2_backward_idx", lambda: (pidx.data_ptr(), dp.data_ptr(), df1.data_ptr(), acc, B, H1, W1, 64, dt),
                             self._elt_meta("maxpool3_bwd", 2 * R1 * 64 + R1 * 64 // 2 + R1 * 64 // 4))
            else:
                self.bwd.add("ksmi_maxpool3x3s2_backward", lambda: (f1.data_ptr(), dp.data_ptr(), df1.data_ptr(), acc, B, H1, W1, 64, dt),
                             self._elt_meta("maxpool3_bwd", 4 * R1 * 64))
            self._bnrelu_bwd("encoder.bn1", df1, f1, s0, sv0, ds0, R1, 64)
            self._linear_bwd("encoder.conv1", col, Kpad, "encoder.conv1.weight", None, ds0, 64, R1, None, k_real=Kreal)
        self._bwd.append(stem_bwd)

        feats = [f1]
        t, cin, h, w = p, 64, H2, W2
        for li, (ch, stride) in enumerate(LAYERS):
            for bi in range(2):
                s_ = stride if bi == 0 else 1
                t = self._basic_block(f"encoder.layer{li + 1}.{bi}", t, cin, ch, h, w, s_)
                h, w, cin = h // s_, w // s_, ch
            feats.append(t)
        self.named.update({f"f{i + 1}": f for i, f in enumerate(feats)})
        skips = feats[::-1]                      # f5, f4, f3, f2, f1
        chans = (512, 256, 128, 64, 64)
        y, cy = skips[0], 512
        for i, co in enumerate(DECODER_CHANNELS):
            skip = skips[i + 1] if i + 1 < len(skips) else None
            cs = chans[i + 1] if skip is not None else 0
            y = self._decoder_block(f"decoder.blocks.{i}", y, cy, skip, cs, co, h, w)
            self.named[f"d{i}"] = y
            h, w, cy = 2 * h, 2 * w, co
        # segmentation head
        P = self.buf(B, H, W, CS)
        nc = self.nc
        self._cv(self.fwd, "segmentation_head", [SrcSpec(y, 16)], [(P, CS, 0, 0, nc, 0)], "segmentation_head.0.weight", H, W, H, W, 3, 1, 1, nc, 16,
                 bias=m._p("segmentation_head.0.bias"))
        HW = H * W
        self.fwd.add("ksmi_out_to_nchw", lambda: (P.data_ptr(), self.logits.data_ptr(), B, nc, CS, HW, 0, dt))

        def head_bwd():
            dP = self.buf(B * HW, CS)
            dy = self.gbuf(y)
            self.gacc(y)
            wk, bk = "segmentation_head.0.weight", "segmentation_head.0.bias"
            self.bwd.add("ksmi_dout_to_nhwc", lambda: (self.dlogits.data_ptr(), self.logits.data_ptr(), dP.data_ptr(), B, nc, CS, HW, 0, dt))
            psrc = [SrcSpec(dP, CS, 0, CS, k_real=nc)]
            self._conv3(self.bwd, "segmentation_head", psrc, [(dy, 16, 0, 0, 16, 0)], wk, None, B, H, W, 16, nc, dgrad=True)
            gview = m._g(wk)[8:]
            self.keep.append(gview)
            dw, ws = make_wgrad(psrc, y, 16, 0, 16, gview, 16 * 9, 9, -1, self._acc_param(wk), B, H, W, H, W, 3, 3, 1, 1, self.dtype)
            self._wgrad(dw, ws, wk)
            rr = max(1, min(512, B * HW // 256))
            self.need("red", rr * CS * 4)
            accb = self._acc_param(bk)
            gb = m._g(bk).data_ptr()
            self.bwd.add("ksmi_channel_sum", lambda: (dP.data_ptr(), self.scr("red"), rr, B * HW, CS, dt), self._elt_meta("channel_sum", B * HW * CS))
            self.bwd.add("ksmi_reduce_rows", lambda: (self.scr("red"), rr, 1, CS, nc, None, None, gb, accb))
            self._mark(bk)
        self._bwd.append(head_bwd)

    # ---------------------------------------------------------------- execution
    def run_forward(self, x):
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.packs.run()
        self.fwd.run()
        return self.logits
