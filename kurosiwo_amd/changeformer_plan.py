"""Static launch plan of one ChangeFormerV6 forward/backward at a fixed (B, H, W, dtype, mode).

Reference computation: /root/reference/models/changeformer.py -- EncoderTransformer_v3.forward_features :430-465 (shared by the
two dates: run here once over 2B images), Block :245-248, Attention :186-208, Mlp :121-133, DecoderTransformer_v3.forward
:568-641, ResidualBlock :478-483, ChangeFormerV6.forward :666-676.

Layouts: token tensors are [rows = 2B*H_s*W_s][C] = NHWC; decoder tensors are NHWC over the B tiles.
"""
import ctypes as C
import os

import torch

from . import _lib
from .changeformer import DEPTHS, EMBED_DIMS, NUM_HEADS, SR_RATIOS
from .plan_base import PlanBase
from .runtime import SrcSpec, conv_grid_m, conv_stats_rows, make_conv, make_wgrad
from .snunet_plan import _Saved

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
CS = 8            # channel stride of the 3-channel NHWC heads (vector-aligned pad channels)
# random-stream sites of one encoder block (site id = 8 * global block index + one of these; oracle/rng_ref.py mirrors them)
SITE_ATTN, SITE_PROJ, SITE_MLP1, SITE_MLP2, SITE_PATH_ATTN, SITE_PATH_MLP = range(6)
NO_SITE = (0, 1.0, 0)


def drop_threshold(p):
    """(thr, inv_keep): an element is dropped when its 32-bit draw < thr = round(p * 2^32); kept ones are scaled by 1/(1-p)"""
    if p <= 0.0:
        return 0, 1.0
    if p >= 1.0:
        raise ValueError("drop probability must be < 1")
    return min(0xFFFFFFFF, int(round(p * 4294967296.0))), 1.0 / (1.0 - p)


class ChangeFormerPlan(PlanBase):
    input_names = ("x1", "x2")
    # the encoder's nn.Linear / sr-conv weight gradients on the train step's side stream (plan_base.PlanBase.side_tokens; waits in
    # _encoder_stage_bwd).  KSMI_CF_SIDE_TOKENS=0: the single-stream list.
    side_tokens = os.environ.get("KSMI_CF_SIDE_TOKENS", "1") != "0"
    slab_bias_side = True      # (plan_base._linear_wgrad: +1.1 % here, profiles/r05_ab_slab_bias.txt)

    def __init__(self, model, B, H, W, dtype, training, with_backward):
        self._init_base(model, dtype, with_backward)
        if H % 32 or W % 32:
            raise ValueError("H and W must be multiples of 32")
        if (H // 32) * (W // 32) != 49:
            raise _lib.KsmiError("ChangeFormerV6 (HIP): the attention kernel is specialised for 224x224 tiles (7x7 = 49 reduced keys)")
        self.B, self.H, self.W, self.training = B, H, W, training
        self.E, self.nc, self.cin = model.embedding_dim, model.output_nc, model.input_nc
        self.x = torch.empty((2 * B, self.cin, H, W), dtype=torch.float32, device=self.dev)
        self.xA, self.xB = self.x[:B], self.x[B:]
        hs = [(H // 32, W // 32), (H // 16, W // 16), (H // 8, W // 8), (H // 4, W // 4), (H, W)]
        self.outputs = [torch.empty((B, self.nc, h, w), dtype=torch.float32, device=self.dev) for h, w in hs]
        self.logits = self.outputs[-1]
        self.dlogits = torch.empty_like(self.logits) if with_backward else None
        self.const = torch.zeros((2, max(self.E, 8)), dtype=torch.float32, device=self.dev)   # row 0 zeros, row 1 ones
        self.const[1].fill_(1.0)
        self._bsteps = []
        # stochastic layers (changeformer.py:267-275: drop_rate = attn_drop_rate = drop_path_rate = 0.1, dpr = linspace(0, 0.1, 13));
        # train mode only, exactly as nn.Dropout / DropPath
        nblk = sum(DEPTHS)
        self.p_drop = float(model.drop_rate) if training else 0.0
        self.fuse_drop = os.environ.get("KSMI_CF_FUSE_DROP", "1") != "0"     # Mlp.drop inside the depth-wise / gelu' passes (round 6)
        self.p_attn = float(model.attn_drop) if training else 0.0
        dp = float(model.drop_path_rate) if training else 0.0
        self.dpr = [dp * i / (nblk - 1) for i in range(nblk)]
        self.stochastic = self.p_drop > 0 or self.p_attn > 0 or dp > 0
        self.rng_ptr = model.rng_state().data_ptr() if self.stochastic else None
        if self.stochastic:
            self.fwd.add("ksmi_rng_advance", lambda: (self.rng_ptr,))
        self._build()
        if with_backward:
            self._build_backward()
        self._finish()

    # ---------------------------------------------------------------- helpers
    def _site(self, gi, site, p):
        thr, inv = drop_threshold(p)
        return (thr, inv, 8 * gi + site)

    def _branch_active(self, gi):
        """does the residual branch of block gi pass through Dropout / DropPath (-> unfused residual add)?"""
        return self.p_drop > 0 or self.dpr[gi] > 0

    def _drop(self, ll, x, resid, y, rows, cols, rows_per_sample, el, path):
        """y = [resid +] x * Dropout(el) * DropPath(path)   (forward of a residual branch; with resid None its backward)"""
        ll.add("ksmi_dropout_apply", lambda: (x.data_ptr(), None if resid is None else resid.data_ptr(), y.data_ptr(), rows, cols, rows_per_sample,
                                              el[0], el[1], el[2], path[0], path[1], path[2], self.rng_ptr, self.dt),
               self._elt_meta("dropout", (2 if resid is None else 3) * rows * cols))

    def _stats_ptr(self):
        return (lambda: self.scr("stats")) if self.training else (lambda: None)

    def _conv3(self, ll, name, srcs, dsts, wkey, bkey, B, H, W, N, Ktot, relu_out=0, alpha=0.0, resid=None, stats=False,
               mask=None, dgrad=False, tag=None):
        """3x3 s1 p1 convolution.  dgrad=True packs W for the input gradient (K = output channels, flipped taps)."""
        d, table = make_conv(srcs, dsts, dsts[0][0], self.m._p(bkey) if bkey else None, None, B, H, W, H, W, 3, 3, 1, 1, N, self.dtype,
                             mask=mask, alpha=alpha, relu_out=relu_out, resid=resid)
        if dgrad:      # element (k = n_out, tap', col = c_in) = W[n][c][flip(tap')]
            d.wpk = self._packed(wkey, table, 9, N, N, N * 9, 9, 0, 1, 1).data_ptr()
        else:          # element (k = c_in, tap, col = n_out) = W[n][c][tap]
            d.wpk = self._packed(wkey, table, 9, N, N, 9, Ktot * 9, 0, 1, 0).data_ptr()
        # statistics rows of the kernel that will run this descriptor (one per persistent workgroup on igemm3 / igemm4, one per M-tile on
        # igemm2; recorded in d.stats_rows: with the tile kernel's count the dispatch kept every convolution WITH statistics off the ring
        # kernel -- diff_c1.0 ran at 660 instead of ~1100 TFLOP/s until round 5)
        rows = conv_stats_rows(d, self.dtype) if stats else conv_grid_m(d)
        if stats:
            self.need("stats", rows * 2 * d.Npad * 4)
            self._later.append(lambda: setattr(d, "stats", self.scr("stats")))
        self._conv(ll, d, tag or ("conv3x3_dgrad" if dgrad else "conv3x3"), name)
        return rows, d.Npad

    def _bn_finalize(self, key, sv, rows, cpad, Cc, count):
        m, tr = self.m, self.training
        g, b = m._p(f"{key}.weight").data_ptr(), m._p(f"{key}.bias").data_ptr()
        rm, rv, nbt = m._b(f"{key}.running_mean").data_ptr(), m._b(f"{key}.running_var").data_ptr(), m._c(f"{key}.num_batches_tracked").data_ptr()
        st = self._stats_ptr()
        self.fwd.add("ksmi_bn_finalize", lambda: (st(), rows, cpad, Cc, float(count), g, b, rm, rv, nbt, BN_MOMENTUM, BN_EPS,
                                                  1 if tr else 0, sv.mean, sv.rstd, sv.scale, sv.shift))

    def _bn_backward(self, key, dy, r, sv, dv, rows, cpad, Cc, count, npix, relu_mask):
        """sums (from the consumer's dgrad epilogue in scratch 'stats') -> dgamma, dbeta, dv"""
        self.need("bnsum", 2 * Cc * 4)
        gw, gb = self.m._g(f"{key}.weight").data_ptr(), self.m._g(f"{key}.bias").data_ptr()
        a1, a2 = self._acc_param(f"{key}.weight"), self._acc_param(f"{key}.bias")
        gamma = self.m._p(f"{key}.weight").data_ptr()
        # the partial rows finish inside the apply pass where the gradient arrives already masked (bnfused.hip MODE 0, as the SNUNet plan
        # since round 4): one launch instead of two; KSMI_BN_FUSED_FAMILIES=0 keeps the separate reduce_rows launch (A/B)
        if relu_mask == 0 and os.environ.get("KSMI_BN_FUSED_FAMILIES", "1") != "0" and self.lib.ksmi_bn_fused_supported(Cc, cpad, self.dt):
            self.bwd.add("ksmi_bn_bwd_fin_apply_gated", lambda: (self.scr("stats"), rows, cpad, self.scr("bnsum"), gw, gb, a1, dy.data_ptr(), r.data_ptr(),
                                                                 sv.mean, sv.rstd, gamma, dv.data_ptr(), float(count), npix, Cc, self.dt),
                         self._elt_meta("bn_bwd_apply", 3 * npix * Cc))
            self._mark(f"{key}.weight", f"{key}.bias")
            return
        self.bwd.add("ksmi_reduce_rows", lambda: (self.scr("stats"), rows, 2, cpad, Cc, self.scr("bnsum"), gw, gb, a1))
        self._mark(f"{key}.weight", f"{key}.bias")
        self.bwd.add("ksmi_bn_bwd_apply", lambda: (dy.data_ptr(), r.data_ptr(), sv.mean, sv.rstd, gamma, self.scr("bnsum"), dv.data_ptr(),
                                                   relu_mask, float(count), npix, Cc, self.dt), self._elt_meta("bn_bwd_apply", 3 * npix * Cc))

    def _nomask(self, r, sv):
        """epilogue 'mask' tuple that only accumulates the BatchNorm-backward sums (sum dy, sum dy*rhat) without masking"""
        return (r, sv.t[0], sv.t[1], self.const[0], self.const[1])

    def _relumask(self, r):
        return (r, self.const[0], self.const[1], self.const[1], self.const[0])

    # ---------------------------------------------------------------- encoder
    def _build(self):
        m, B2, dt = self.m, 2 * self.B, self.dt
        E = self.E
        feats = []
        src, src_c, Hs, Ws = self.x, self.cin, self.H, self.W
        for st in range(4):
            Cc, heads, sr, stride = EMBED_DIMS[st], NUM_HEADS[st], SR_RATIOS[st], (4 if st == 0 else 2)
            Hi, Wi = Hs, Ws
            Hs, Ws = Hi // stride, Wi // stride
            R, Rk = B2 * Hs * Ws, B2 * 49
            pe = f"Tenc_x2.patch_embed{st + 1}"
            self._stage("patch_embed")
            kc = 32 if self.dtype == torch.bfloat16 else 16
            Kreal = src_c * 49
            Kpad = -(-Kreal // kc) * kc
            col, t0, t = self.buf(R, Kpad), self.buf(R, Cc), self.buf(R, Cc)
            pe_tc = st > 0 and self._tc_ok(src_c)
            pe_wtc = None
            if pe_tc:
                self.fwd.add("ksmi_im2col_tc", lambda src=src, col=col, src_c=src_c, Hi=Hi, Wi=Wi, Hs=Hs, Ws=Ws, stride=stride, Kpad=Kpad: (
                    src.data_ptr(), col.data_ptr(), B2, src_c, Hi, Wi, Hs, Ws, 7, 7, stride, 3, Kpad, dt), self._elt_meta("im2col", 2 * R * Kpad))
                pe_wtc = self._linear_tc(f"pe{st + 1}", col, Kpad, f"{pe}.proj.weight", f"{pe}.proj.bias", t0, Cc, R, src_c, 49)
            else:
                self.fwd.add("ksmi_im2col", lambda src=src, col=col, src_c=src_c, Hi=Hi, Wi=Wi, Hs=Hs, Ws=Ws, stride=stride, Kpad=Kpad, st=st: (
                    src.data_ptr(), col.data_ptr(), B2, src_c, Hi, Wi, Hs, Ws, 7, 7, stride, 3, Kpad, 1 if st == 0 else 0, dt),
                    self._elt_meta("im2col", 2 * R * Kpad))
                self._linear(f"pe{st + 1}", col, Kpad, f"{pe}.proj.weight", f"{pe}.proj.bias", t0, Cc, R, k_real=Kreal)
            st_pe = self._ln(t0, f"{pe}.norm.weight", f"{pe}.norm.bias", t, R, Cc, 1e-5)
            self.named[f"pe{st + 1}"] = t
            blocks = []
            ptmp = self.buf(R, Cc) if self.stochastic else None     # branch output before Dropout / DropPath / residual add
            for i in range(DEPTHS[st]):
                k = f"Tenc_x2.block{st + 1}.{i}"
                rec = dict(k=k, t_in=t, gi=sum(DEPTHS[:st]) + i)
                gi = rec["gi"]
                h, q, att, t_mid = self.buf(R, Cc), self.buf(R, Cc), self.buf(R, Cc), self.buf(R, Cc)
                self._stage("attention_block")          # Block.forward's first half: norm1 -> Attention (q, sr conv, norm, kv, softmax(qk^T)v, proj) -> drop + residual
                rec["st1"] = self._ln(t, f"{k}.norm1.weight", f"{k}.norm1.bias", h, R, Cc, 1e-6)
                self._linear(f"{k}.q", h, Cc, f"{k}.attn.q.weight", f"{k}.attn.q.bias", q, Cc, R)
                if sr > 1:
                    Ksr = Cc * sr * sr
                    col2, xr, xn = self.buf(Rk, Ksr), self.buf(Rk, Cc), self.buf(Rk, Cc)
                    if self._tc_ok(Cc):
                        self.fwd.add("ksmi_im2col_tc", lambda h=h, col2=col2, Cc=Cc, Hs=Hs, Ws=Ws, sr=sr, Ksr=Ksr: (
                            h.data_ptr(), col2.data_ptr(), B2, Cc, Hs, Ws, Hs // sr, Ws // sr, sr, sr, sr, 0, Ksr, dt),
                            self._elt_meta("im2col", 2 * Rk * Ksr))
                        rec["sr_wtc"] = self._linear_tc(f"{k}.sr", col2, Ksr, f"{k}.attn.sr.weight", f"{k}.attn.sr.bias", xr, Cc, Rk, Cc, sr * sr)
                    else:
                        self.fwd.add("ksmi_im2col", lambda h=h, col2=col2, Cc=Cc, Hs=Hs, Ws=Ws, sr=sr, Ksr=Ksr: (
                            h.data_ptr(), col2.data_ptr(), B2, Cc, Hs, Ws, Hs // sr, Ws // sr, sr, sr, sr, 0, Ksr, 0, dt),
                            self._elt_meta("im2col", 2 * Rk * Ksr))
                        self._linear(f"{k}.sr", col2, Ksr, f"{k}.attn.sr.weight", f"{k}.attn.sr.bias", xr, Cc, Rk)
                    rec["st_sr"] = self._ln(xr, f"{k}.attn.norm.weight", f"{k}.attn.norm.bias", xn, Rk, Cc, 1e-5)
                    rec.update(col2=col2, xr=xr, xn=xn, Ksr=Ksr)
                else:
                    xn = h
                    rec.update(xn=h)
                kv = self.buf(Rk, 2 * Cc)
                self._linear(f"{k}.kv", xn, Cc, f"{k}.attn.kv.weight", f"{k}.attn.kv.bias", kv, 2 * Cc, Rk)
                scale = float(Cc // heads) ** -0.5
                aflops = 4 * B2 * Hs * Ws * 49 * Cc
                ad = self._site(gi, SITE_ATTN, self.p_attn)
                self.fwd.add("ksmi_sr_attention_forward_drop", lambda q=q, kv=kv, att=att, Hs=Hs, Ws=Ws, heads=heads, Cc=Cc, scale=scale, ad=ad: (
                    q.data_ptr(), kv.data_ptr(), att.data_ptr(), B2, Hs * Ws, 49, heads, Cc, scale, ad[0], ad[1], ad[2], self.rng_ptr, dt),
                    {"kind": "sr_attention_fwd", "bytes": (3 * R * Cc + 2 * Rk * Cc) * self._es(), "flops": aflops})
                if self._branch_active(gi):
                    self._linear(f"{k}.proj", att, Cc, f"{k}.attn.proj.weight", f"{k}.attn.proj.bias", ptmp, Cc, R)
                    self._drop(self.fwd, ptmp, t, t_mid, R, Cc, Hs * Ws, self._site(gi, SITE_PROJ, self.p_drop), self._site(gi, SITE_PATH_ATTN, self.dpr[gi]))
                else:
                    self._linear(f"{k}.proj", att, Cc, f"{k}.attn.proj.weight", f"{k}.attn.proj.bias", t_mid, Cc, R, resid=t)
                h2, u, z, g, t_out = self.buf(R, Cc), self.buf(R, 4 * Cc), self.buf(R, 4 * Cc), self.buf(R, 4 * Cc), self.buf(R, Cc)
                self._stage("mlp_block")
                rec["st2"] = self._ln(t_mid, f"{k}.norm2.weight", f"{k}.norm2.bias", h2, R, Cc, 1e-6)
                self._linear(f"{k}.fc1", h2, Cc, f"{k}.mlp.fc1.weight", f"{k}.mlp.fc1.bias", u, 4 * Cc, R)
                wd, bd = m._p(f"{k}.mlp.dwconv.dwconv.weight").data_ptr(), m._p(f"{k}.mlp.dwconv.dwconv.bias").data_ptr()
                # Mlp.drop after the activation (:130) rides on the depth-wise kernel's store of g (round 6; fc2 and its weight gradient read
                # the dropped g): one launch and two passes over the 4C-wide tensor less per block; KSMI_CF_FUSE_DROP=0: the separate pass
                fuse_d = self.fuse_drop and self._branch_active(gi) and self.p_drop > 0
                if fuse_d:
                    ds = self._site(gi, SITE_MLP1, self.p_drop)
                    self.fwd.add("ksmi_dwconv3x3_gelu_forward_drop", lambda u=u, z=z, g=g, wd=wd, bd=bd, Hs=Hs, Ws=Ws, Cc=Cc, ds=ds: (
                        u.data_ptr(), wd, bd, z.data_ptr(), g.data_ptr(), B2, Hs, Ws, 4 * Cc, ds[0], ds[1], ds[2], self.rng_ptr, dt),
                        self._elt_meta("dwconv_gelu", 3 * R * 4 * Cc))
                else:
                    self.fwd.add("ksmi_dwconv3x3_gelu_forward", lambda u=u, z=z, g=g, wd=wd, bd=bd, Hs=Hs, Ws=Ws, Cc=Cc: (
                        u.data_ptr(), wd, bd, z.data_ptr(), g.data_ptr(), B2, Hs, Ws, 4 * Cc, dt), self._elt_meta("dwconv_gelu", 3 * R * 4 * Cc))
                if self._branch_active(gi):
                    if self.p_drop > 0 and not fuse_d:             # Mlp.drop after the activation (:130): in place, fc2 and its wgrad read it
                        self._drop(self.fwd, g, None, g, R, 4 * Cc, Hs * Ws, self._site(gi, SITE_MLP1, self.p_drop), NO_SITE)
                    self._linear(f"{k}.fc2", g, 4 * Cc, f"{k}.mlp.fc2.weight", f"{k}.mlp.fc2.bias", ptmp, Cc, R)
                    self._drop(self.fwd, ptmp, t_mid, t_out, R, Cc, Hs * Ws, self._site(gi, SITE_MLP2, self.p_drop), self._site(gi, SITE_PATH_MLP, self.dpr[gi]))
                else:
                    self._linear(f"{k}.fc2", g, 4 * Cc, f"{k}.mlp.fc2.weight", f"{k}.mlp.fc2.bias", t_out, Cc, R, resid=t_mid)
                rec.update(h=h, q=q, kv=kv, att=att, t_mid=t_mid, h2=h2, u=u, z=z, g=g, scale=scale, aflops=aflops)
                blocks.append(rec)
                t = t_out
                self.named[f"s{st + 1}b{i}"] = t
            f = self.buf(R, Cc)
            self._stage("patch_embed")
            st_n = self._ln(t, f"Tenc_x2.norm{st + 1}.weight", f"Tenc_x2.norm{st + 1}.bias", f, R, Cc, 1e-6)
            self.named[f"f{st + 1}"] = f
            feats.append(dict(f=f, C=Cc, H=Hs, W=Ws, R=R, Rk=Rk, heads=heads, sr=sr, stride=stride, Hi=Hi, Wi=Wi, src=src, src_c=src_c,
                              col=col, Kpad=Kpad, Kreal=Kreal, t0=t0, st_pe=st_pe, pe=pe, blocks=blocks, t_last=t, st_n=st_n, st=st, pe_wtc=pe_wtc))
            src, src_c = f, Cc
        self.feats = feats
        self._stage("decoder")
        self._build_decoder()

    # ---------------------------------------------------------------- decoder (changeformer.py:568-641)
    def _build_decoder(self):
        m, B, E, dt, nc = self.m, self.B, self.E, self.dt, self.nc
        D = "TDec_x2"
        H1, W1 = self.feats[0]["H"], self.feats[0]["W"]
        scales, prev = {}, None
        for i in (4, 3, 2, 1):
            ft = self.feats[i - 1]
            h, w, Ci, R = ft["H"], ft["W"], ft["C"], ft["R"]
            npix = B * h * w
            L = self.buf(R, E)                                    # [2B, h, w, E]: first B tiles = date 1, last B = date 2
            self._linear(f"linear_c{i}", ft["f"], Ci, f"{D}.linear_c{i}.proj.weight", f"{D}.linear_c{i}.proj.bias", L, E, R)
            LA, LB = L[:npix], L[npix:]
            r1, r2 = self.buf(npix, E), self.buf(npix, E)
            sv = _Saved(E, self.dev)
            rows, cpad = self._conv3(self.fwd, f"diff_c{i}.0", [SrcSpec(LA, E), SrcSpec(LB, E)], [(r1, E, 0, 0, E, 0)], f"{D}.diff_c{i}.0.weight",
                                     f"{D}.diff_c{i}.0.bias", B, h, w, E, 2 * E, relu_out=1, stats=self.training)
            self._bn_finalize(f"{D}.diff_c{i}.2", sv, rows, cpad, E, npix)
            self._conv3(self.fwd, f"diff_c{i}.3", [SrcSpec(r1, E, scale=sv.scale_t, shift=sv.shift_t, relu=0)], [(r2, E, 0, 0, E, 0)],
                        f"{D}.diff_c{i}.3.weight", f"{D}.diff_c{i}.3.bias", B, h, w, E, E, relu_out=1)
            if prev is None:
                c = r2
            else:
                c = self.buf(npix, E)
                pc, ph, pw = prev
                self.fwd.add("ksmi_bilinear_forward", lambda pc=pc, r2=r2, c=c, ph=ph, pw=pw, h=h, w=w: (
                    pc.data_ptr(), r2.data_ptr(), c.data_ptr(), B, ph, pw, h, w, E, dt), self._elt_meta("bilinear", 3 * npix * E))
            self.named[f"c{i}"] = c
            # make_prediction head (side output)
            m1, pr = self.buf(npix, CS), self.buf(npix, CS)
            sv3 = _Saved(16, self.dev)
            rows3, cpad3 = self._conv3(self.fwd, f"make_pred_c{i}.0", [SrcSpec(c, E)], [(m1, CS, 0, 0, nc, 0)], f"{D}.make_pred_c{i}.0.weight",
                                       f"{D}.make_pred_c{i}.0.bias", B, h, w, nc, E, relu_out=1, stats=self.training)
            self._bn_finalize(f"{D}.make_pred_c{i}.2", sv3, rows3, cpad3, nc, npix)
            self._conv3(self.fwd, f"make_pred_c{i}.3", [SrcSpec(m1, CS, 0, CS, scale=sv3.scale_t, shift=sv3.shift_t, relu=0, k_real=nc)],
                        [(pr, CS, 0, 0, nc, 0)], f"{D}.make_pred_c{i}.3.weight", f"{D}.make_pred_c{i}.3.bias", B, h, w, nc, nc)
            out = self.outputs[4 - i]
            act = 1 if m.decoder_softmax else 0
            self.fwd.add("ksmi_out_to_nchw", lambda pr=pr, out=out, h=h, w=w: (pr.data_ptr(), out.data_ptr(), B, nc, CS, h * w, act, dt))
            if i == 1:
                up = c
            else:
                up = self.buf(B * H1 * W1, E)
                self.fwd.add("ksmi_bilinear_forward", lambda c=c, up=up, h=h, w=w: (c.data_ptr(), None, up.data_ptr(), B, h, w, H1, W1, E, dt),
                             self._elt_meta("bilinear", 2 * B * H1 * W1 * E))
            scales[i] = dict(L=L, LA=LA, LB=LB, r1=r1, r2=r2, sv=sv, c=c, up=up, h=h, w=w, npix=npix, ft=ft, prev=prev)
            prev = (c, h, w)
        self.scales = scales
        # linear_fuse: Conv1x1(4E -> E) over cat(_c4_up, _c3_up, _c2_up, _c1) + BatchNorm
        np1 = B * H1 * W1
        F0, Fb = self.buf(np1, E), self.buf(np1, E)
        svF = _Saved(E, self.dev)
        fsrcs = [SrcSpec(scales[i]["up"], E) for i in (4, 3, 2, 1)]
        d, table = make_conv(fsrcs, [(F0, E, 0, 0, E, 0)], F0, m._p(f"{D}.linear_fuse.0.bias"), None, B, H1, W1, H1, W1, 1, 1, 1, 0, E, self.dtype)
        d.wpk = self._packed(f"{D}.linear_fuse.0.weight", table, 1, E, E, 1, 4 * E, 0, 0).data_ptr()
        rowsF = conv_stats_rows(d, self.dtype) if self.training else conv_grid_m(d)   # (rows of the kernel that will run it: see changeformer_plan._conv3)
        if self.training:
            self.need("stats", rowsF * 2 * d.Npad * 4)
            self._later.append(lambda: setattr(d, "stats", self.scr("stats")))
        self._conv(self.fwd, d, "conv1x1", "linear_fuse.0")
        self._bn_finalize(f"{D}.linear_fuse.1", svF, rowsF, d.Npad, E, np1)
        self.fwd.add("ksmi_affine", lambda: (F0.data_ptr(), svF.scale, svF.shift, Fb.data_ptr(), np1, E, 0, C.c_float(1.0), dt),
                     self._elt_meta("bn_apply", 2 * np1 * E))
        self.named["fuse"] = Fb
        # convd2x -> dense_2x -> convd1x -> dense_1x -> change_probability
        X2, Ra, Y2 = self.buf(B, 2 * H1, 2 * W1, E), self.buf(B, 2 * H1, 2 * W1, E), self.buf(B, 2 * H1, 2 * W1, E)
        X1, Rb, Y1 = self.buf(B, 4 * H1, 4 * W1, E), self.buf(B, 4 * H1, 4 * W1, E), self.buf(B, 4 * H1, 4 * W1, E)
        P = self.buf(B, 4 * H1, 4 * W1, CS)
        self._deconv("convd2x", Fb, E, E, H1, W1, X2, E, prefix=f"{D}.", suffix=".conv2d")
        self._res_block("dense_2x.0", X2, Ra, Y2, 2 * H1, 2 * W1)
        self._deconv("convd1x", Y2, E, E, 2 * H1, 2 * W1, X1, E, prefix=f"{D}.", suffix=".conv2d")
        self._res_block("dense_1x.0", X1, Rb, Y1, 4 * H1, 4 * W1)
        self._conv3(self.fwd, "change_probability", [SrcSpec(Y1, E)], [(P, CS, 0, 0, nc, 0)], f"{D}.change_probability.conv2d.weight",
                    f"{D}.change_probability.conv2d.bias", B, 4 * H1, 4 * W1, nc, E)
        act = 1 if m.decoder_softmax else 0
        HW = 16 * H1 * W1
        self.fwd.add("ksmi_out_to_nchw", lambda: (P.data_ptr(), self.logits.data_ptr(), B, nc, CS, HW, act, dt))
        self.named.update(dense_2x=Y2, dense_1x=Y1)
        self.dec = dict(F0=F0, Fb=Fb, svF=svF, rowsF=rowsF, X2=X2, Ra=Ra, Y2=Y2, X1=X1, Rb=Rb, Y1=Y1, P=P, H1=H1, W1=W1, np1=np1, fsrcs=fsrcs)

    def _res_block(self, name, X, R_, Y, H, W):
        """ResidualBlock (:471-483): Y = 0.1 * conv2(relu(conv1(X))) + X"""
        D, E, B = "TDec_x2", self.E, self.B
        self._conv3(self.fwd, f"{name}.conv1", [SrcSpec(X, E)], [(R_, E, 0, 0, E, 0)], f"{D}.{name}.conv1.conv2d.weight",
                    f"{D}.{name}.conv1.conv2d.bias", B, H, W, E, E, relu_out=1)
        self._conv3(self.fwd, f"{name}.conv2", [SrcSpec(R_, E)], [(Y, E, 0, 0, E, 0)], f"{D}.{name}.conv2.conv2d.weight",
                    f"{D}.{name}.conv2.conv2d.bias", B, H, W, E, E, alpha=0.1, resid=(X, E))

    # ================================================================ backward
    def _wgrad3(self, srcs, dy, dyC, N, wkey, B, H, W, Ktot):
        dw, ws = make_wgrad(srcs, dy, dyC, 0, N, self.m._g(wkey), 9, Ktot * 9, 1, self._acc_param(wkey), B, H, W, H, W, 3, 3, 1, 1, self.dtype)
        self._wgrad(dw, ws, wkey)

    def _res_block_bwd(self, name, X, R_, dY, H, W):
        """in: dY (gradient of the block output, buffer reused as dX on return)"""
        D, E, B = "TDec_x2", self.E, self.B
        npix = B * H * W
        g2, dR = self.buf(npix, E), self.buf(npix, E)
        dt = self.dt
        self.bwd.add("ksmi_affine", lambda: (dY.data_ptr(), None, None, g2.data_ptr(), npix, E, 0, C.c_float(0.1), dt),
                     self._elt_meta("scale", 2 * npix * E))
        w2, b2 = f"{D}.{name}.conv2.conv2d.weight", f"{D}.{name}.conv2.conv2d.bias"
        w1, b1 = f"{D}.{name}.conv1.conv2d.weight", f"{D}.{name}.conv1.conv2d.bias"
        self._conv3(self.bwd, f"{name}.conv2", [SrcSpec(g2, E)], [(dR, E, 0, 0, E, 0)], w2, None, B, H, W, E, E, mask=self._relumask(R_), dgrad=True)
        self._wgrad3([SrcSpec(R_, E)], g2, E, E, w2, B, H, W, E)
        self._bias_grad(g2, npix, E, b2)
        self._conv3(self.bwd, f"{name}.conv1", [SrcSpec(dR, E)], [(dY, E, 0, 0, E, 1)], w1, None, B, H, W, E, E, dgrad=True)
        self._wgrad3([SrcSpec(X, E)], dR, E, E, w1, B, H, W, E)
        self._bias_grad(dR, npix, E, b1)

    def _build_backward(self):
        m, B, E, dt, nc = self.m, self.B, self.E, self.dt, self.nc
        self._stage("decoder")
        D = "TDec_x2"
        dec = self.dec
        H1, W1, np1 = dec["H1"], dec["W1"], dec["np1"]
        HW = 16 * H1 * W1
        act = 1 if m.decoder_softmax else 0
        # the gradient of the 3-channel map is kept at a 32-channel stride (zero pad channels): one whole 64-byte k-chunk per pixel
        # puts its 3x3 input-gradient convolution on the pipelined kernel (K = 8 ran on the v1 kernel: 1.35 ms vs 0.3)
        CSB = 32
        dP = self.buf(B * HW, CSB)
        dY1, dY2 = self.buf(B * HW, E), self.buf(B * 4 * H1 * W1, E)
        self.bwd.add("ksmi_dout_to_nhwc", lambda: (self.dlogits.data_ptr(), self.logits.data_ptr(), dP.data_ptr(), B, nc, CSB, HW, act, dt))
        # change_probability: dY1 = conv^T(dP); dW via the operand swap (halo side = dP): G[tap][o][c] = dW[o][c][8 - tap]
        wk, bk = f"{D}.change_probability.conv2d.weight", f"{D}.change_probability.conv2d.bias"
        psrc = [SrcSpec(dP, CSB, 0, CSB, k_real=nc)]
        self._conv3(self.bwd, "change_probability", psrc, [(dY1, E, 0, 0, E, 0)], wk, None, B, 4 * H1, 4 * W1, E, nc, dgrad=True)
        gview = m._g(wk)[8:]
        dw, ws = make_wgrad(psrc, dec["Y1"], E, 0, E, gview, E * 9, 9, -1, self._acc_param(wk), B, 4 * H1, 4 * W1, 4 * H1, 4 * W1, 3, 3, 1, 1, self.dtype)
        self.keep.append(gview)
        self._wgrad(dw, ws, wk)
        rr = max(1, min(self.csum_rows, B * HW // 256))
        slot = self._rs_slot(rr * CSB * 4)
        self.bwd.add("ksmi_channel_sum", lambda: (dP.data_ptr(), self.scr(slot), rr, B * HW, CSB, dt), self._elt_meta("channel_sum", B * HW * CSB))
        self._defer_rowsum(bk, slot, 0, rr, 1, 0, CSB, nc)
        self._rs_tick()
        # dense_1x, convd1x, dense_2x, convd2x
        self._res_block_bwd("dense_1x.0", dec["X1"], dec["Rb"], dY1, 4 * H1, 4 * W1)            # dY1 now holds dX1
        self._deconv_bwd("convd1x", dec["Y2"], E, E, 2 * H1, 2 * W1, dY1, E, dY2, prefix=f"{D}.", suffix=".conv2d")
        self._res_block_bwd("dense_2x.0", dec["X2"], dec["Ra"], dY2, 2 * H1, 2 * W1)            # dY2 now holds dX2
        dFb, dF0 = self.buf(np1, E), self.buf(np1, E)
        self._deconv_bwd_stats("convd2x", dec["Fb"], E, E, H1, W1, dY2, dFb, dec["F0"], dec["svF"])
        # linear_fuse
        self._bn_backward(f"{D}.linear_fuse.1", dFb, dec["F0"], dec["svF"], dF0, self._last_rows, self._last_cpad, E, np1, np1, 0)
        ups = {i: self.scales[i]["up"] for i in (4, 3, 2, 1)}
        dups = {i: self.buf(np1, E) for i in (4, 3, 2, 1)}
        wf, bf = f"{D}.linear_fuse.0.weight", f"{D}.linear_fuse.0.bias"
        dsts = [(dups[i], E, 0, j * E, E, 0) for j, i in enumerate((4, 3, 2, 1))]
        d, table = make_conv([SrcSpec(dF0, E)], dsts, dups[4], None, None, B, H1, W1, H1, W1, 1, 1, 1, 0, 4 * E, self.dtype)
        d.wpk = self._packed(wf, table, 1, 4 * E, 4 * E, 4 * E, 1, 0, 0).data_ptr()
        self._conv(self.bwd, d, "conv1x1_dgrad", "linear_fuse.0")
        dw, ws = make_wgrad(dec["fsrcs"], dF0, E, 0, E, m._g(wf), 1, 4 * E, 0, self._acc_param(wf), B, H1, W1, H1, W1, 1, 1, 1, 0, self.dtype)
        self._wgrad(dw, ws, wf)
        self._bias_grad(dF0, np1, E, bf)
        # scales 1 -> 4
        dc_prev = None
        for i in (1, 2, 3, 4):
            sc = self.scales[i]
            h, w, npix, ft = sc["h"], sc["w"], sc["npix"], sc["ft"]
            if i == 1:
                dc = dups[1]
            else:
                dc = self.buf(npix, E)
                self.bwd.add("ksmi_bilinear_backward", lambda du=dups[i], dc=dc, h=h, w=w: (du.data_ptr(), dc.data_ptr(), 0, B, h, w, H1, W1, E, dt),
                             self._elt_meta("bilinear_bwd", 2 * np1 * E))
                pdc, ph, pw = dc_prev
                self.bwd.add("ksmi_bilinear_backward", lambda pdc=pdc, dc=dc, h=h, w=w, ph=ph, pw=pw: (pdc.data_ptr(), dc.data_ptr(), 1, B, h, w, ph, pw, E, dt),
                             self._elt_meta("bilinear_bwd", 2 * B * ph * pw * E))
            dc_prev = (dc, h, w)
            r1, r2, sv = sc["r1"], sc["r2"], sc["sv"]
            dv2, dy1, dv1 = self.buf(npix, E), self.buf(npix, E), self.buf(npix, E)
            self.bwd.add("ksmi_relu_backward", lambda dc=dc, r2=r2, dv2=dv2, npix=npix: (dc.data_ptr(), r2.data_ptr(), dv2.data_ptr(), npix * E, dt),
                         self._elt_meta("relu_bwd", 3 * npix * E))
            w3, b3 = f"{D}.diff_c{i}.3.weight", f"{D}.diff_c{i}.3.bias"
            w0, b0 = f"{D}.diff_c{i}.0.weight", f"{D}.diff_c{i}.0.bias"
            rows, cpad = self._conv3(self.bwd, f"diff_c{i}.3", [SrcSpec(dv2, E)], [(dy1, E, 0, 0, E, 0)], w3, None, B, h, w, E, E,
                                     mask=self._nomask(r1, sv), stats=True, dgrad=True)
            self._wgrad3([SrcSpec(r1, E, scale=sv.scale_t, shift=sv.shift_t, relu=0)], dv2, E, E, w3, B, h, w, E)
            self._bias_grad(dv2, npix, E, b3)
            self._bn_backward(f"{D}.diff_c{i}.2", dy1, r1, sv, dv1, rows, cpad, E, npix, npix, 1)
            dL = self.buf(ft["R"], E)
            self._conv3(self.bwd, f"diff_c{i}.0", [SrcSpec(dv1, E)], [(dL[:npix], E, 0, 0, E, 0), (dL[npix:], E, 0, E, E, 0)], w0, None,
                        B, h, w, 2 * E, E, dgrad=True)
            self._wgrad3([SrcSpec(sc["LA"], E), SrcSpec(sc["LB"], E)], dv1, E, E, w0, B, h, w, 2 * E)
            self._bias_grad(dv1, npix, E, b0)
            df = self.buf(ft["R"], ft["C"])
            self._linear_bwd(f"linear_c{i}", ft["f"], ft["C"], f"{D}.linear_c{i}.proj.weight", f"{D}.linear_c{i}.proj.bias", dL, E, ft["R"], df)
            ft["df"] = df
        # make_prediction heads receive no gradient (multi_scale_train false): their gradients are exact zeros
        for i in (4, 3, 2, 1):
            for sfx in ("0.weight", "0.bias", "2.weight", "2.bias", "3.weight", "3.bias"):
                self._zero_grad_key(f"{D}.make_pred_c{i}.{sfx}")
        for st in (3, 2, 1, 0):
            self._encoder_stage_bwd(self.feats[st])

    def _zero_grad_key(self, key):
        g = self.m._g(key)
        self._pinit.add(key)
        self.bwd.add("ksmi_fill_zero", lambda: (g.data_ptr(), g.numel() * 4))
        self._mark(key)

    def _deconv_bwd_stats(self, name, x, Cin, N, H, W, dout, dx, r, sv):
        """_deconv_bwd whose input-gradient epilogue also accumulates the BatchNorm-backward sums of the deconv input"""
        D, B = "TDec_x2", self.B
        wkey, bkey = f"{D}.{name}.conv2d.weight", f"{D}.{name}.conv2d.bias"
        src = [SrcSpec(dout, N)]
        # four 2x2 phase convolutions over the parity sub-images of dout (see PlanBase._deconv_bwd); the BatchNorm-backward sums are
        # linear in the gradient, so every phase writes its own block of partial rows and reduce_rows walks all of them
        rows_total, first = 0, True
        for py in range(2):
            for px in range(2):
                tap_map = []
                for a in range(2):
                    for b in range(2):
                        tap_map.append((2 * a if py else 1 + 2 * a) * 4 + (2 * b if px else 1 + 2 * b))
                d, table = make_conv(src, [(dx, Cin, 0, 0, Cin, 0 if first else 1)], dx, None, None, B, H, W, H, W, 2, 2, 1, py, Cin, self.dtype,
                                     mask=self._nomask(r, sv), pad_x=px, in_map=(2, 2, py, px, 2 * H, 2 * W))
                d.wpk = self._packed(wkey, table, 4, Cin, Cin, 16, N * 16, 0, 1, 0, tap_map).data_ptr()
                rows = conv_stats_rows(d, self.dtype)               # (one row per persistent workgroup on igemm4's 2 x 2 instance, per M-tile on igemm2)
                off = rows_total * 2 * d.Npad * 4
                rows_total += rows
                self._later.append(lambda d=d, off=off: setattr(d, "stats", self.scr("stats") + off))
                self._conv(self.bwd, d, "deconv_dgrad_phase", f"{name}.p{py}{px}")
                first = False
        self.need("stats", rows_total * 2 * d.Npad * 4)
        self._last_rows, self._last_cpad = rows_total, d.Npad
        self._deconv_wgrad(src, x, Cin, N, H, W, wkey, B)
        self._bias_grad(dout, B * 4 * H * W, N, bkey)

    def _encoder_stage_bwd(self, ft):
        m, B2, dt = self.m, 2 * self.B, self.dt
        Cc, Hs, Ws, R, Rk, heads, sr = ft["C"], ft["H"], ft["W"], ft["R"], ft["Rk"], ft["heads"], ft["sr"]
        gt = self.buf(R, Cc)
        tC, t4, tq, tkv, tk = self.buf(R, Cc), self.buf(R, 4 * Cc), self.buf(R, Cc), self.buf(Rk, 2 * Cc), self.buf(Rk, Cc)
        st = ft["st"]
        self._stage("patch_embed")
        self._ln_bwd(ft["df"], ft["t_last"], ft["st_n"], f"Tenc_x2.norm{st + 1}.weight", f"Tenc_x2.norm{st + 1}.bias", gt, 0, R, Cc)
        ws_attn = self.lib.ksmi_sr_attention_bwd_workspace(B2, Hs * Ws, 49, heads, Cc)
        self.need("attn", ws_attn)
        N = Hs * Ws
        tD = self.buf(R, Cc) if self.stochastic else None           # gradient of a branch output behind Dropout / DropPath
        # Side-stream weight gradients (side_tokens): fc2 / fc1 / proj / kv / sr / q of a block may run next to the rest of its
        # backward pass.  Their operands are either per-block forward activations (live until the next forward) or the stage buffers
        # tD / tDa / tq / tkv / dxr, which the NEXT block overwrites: a tagged wait stands before each of those writes.  A gradient
        # whose dY is gt itself (block without Dropout / DropPath: gt is accumulated into further down the block) stays in line.
        S = self.side_tokens
        tDa = self.buf(R, Cc) if (self.stochastic and S) else tD    # own buffer for the attention branch: proj's dY outlives the Mlp's
        dxr_s = self.buf(Rk, Cc) if S else None
        dh_s = self.buf(R, Cc) if S else None
        prev = None
        for rec in reversed(ft["blocks"]):
            k, gi = rec["k"], rec["gi"]
            active = self._branch_active(gi)
            sd = S and active
            # Mlp
            self._stage("mlp_block")
            gy = gt
            if active:
                if S and prev:
                    self.bwd.add_wait_side(f"{prev}.fc2")
                self._drop(self.bwd, gt, None, tD, R, Cc, N, self._site(gi, SITE_MLP2, self.p_drop), self._site(gi, SITE_PATH_MLP, self.dpr[gi]))
                gy = tD
            self._linear_bwd(f"{k}.fc2", rec["g"], 4 * Cc, f"{k}.mlp.fc2.weight", f"{k}.mlp.fc2.bias", gy, Cc, R, t4,
                             side_tag=f"{k}.fc2" if sd else None)
            if active and self.p_drop > 0 and self.fuse_drop:       # the mask of Mlp.drop and gelu' in one pass over the gradient (round 6)
                ds = self._site(gi, SITE_MLP1, self.p_drop)
                self.bwd.add("ksmi_gelu_backward_drop", lambda z=rec["z"], ds=ds: (t4.data_ptr(), z.data_ptr(), t4.data_ptr(), R * 4 * Cc,
                                                                                 ds[0], ds[1], ds[2], self.rng_ptr, dt),
                             self._elt_meta("gelu_bwd", 3 * R * 4 * Cc))
            else:
                if active and self.p_drop > 0:
                    self._drop(self.bwd, t4, None, t4, R, 4 * Cc, N, self._site(gi, SITE_MLP1, self.p_drop), NO_SITE)
                self.bwd.add("ksmi_gelu_backward", lambda z=rec["z"]: (t4.data_ptr(), z.data_ptr(), t4.data_ptr(), R * 4 * Cc, dt),
                             self._elt_meta("gelu_bwd", 3 * R * 4 * Cc))
            wd = m._p(f"{k}.mlp.dwconv.dwconv.weight").data_ptr()
            du = rec["z"]                                         # z is dead after gelu_backward: reuse as d(fc1 output)
            self.bwd.add("ksmi_dwconv3x3_backward_input", lambda du=du, wd=wd: (t4.data_ptr(), wd, du.data_ptr(), B2, Hs, Ws, 4 * Cc, dt),
                         self._elt_meta("dwconv_bwd", 2 * R * 4 * Cc))
            rows = max(1, min(1024 // max(1, -(-(4 * Cc // (8 if self.dtype == torch.bfloat16 else 4)) // 64)), R // 16))
            kw, kb = f"{k}.mlp.dwconv.dwconv.weight", f"{k}.mlp.dwconv.dwconv.bias"
            C4 = 4 * Cc
            slot = self._rs_slot(rows * 10 * C4 * 4)
            self.bwd.add("ksmi_dwconv3x3_wgrad", lambda u=rec["u"], rows=rows, slot=slot: (u.data_ptr(), t4.data_ptr(), self.scr(slot), rows, B2, Hs, Ws, C4, dt),
                         self._elt_meta("dwconv_wgrad", 2 * R * C4))
            self._defer_rowsum(kw, slot, 0, rows, 1, 0, 10 * C4, 9 * C4)           # partial rows [rows][9*C4 | C4]
            self._defer_rowsum(kb, slot, 9 * C4 * 4, rows, 1, 0, 10 * C4, C4)
            self._rs_tick()
            self._linear_bwd(f"{k}.fc1", rec["h2"], Cc, f"{k}.mlp.fc1.weight", f"{k}.mlp.fc1.bias", du, 4 * Cc, R, tC,
                             side_tag=f"{k}.fc1" if S else None)       # h2 and z belong to this block: nothing overwrites them (dh_s below)
            self._ln_bwd(tC, rec["t_mid"], rec["st2"], f"{k}.norm2.weight", f"{k}.norm2.bias", gt, 1, R, Cc)
            # Attention
            self._stage("attention_block")
            gy = gt
            if active:
                if S and prev:
                    self.bwd.add_wait_side(f"{prev}.proj")
                self._drop(self.bwd, gt, None, tDa, R, Cc, N, self._site(gi, SITE_PROJ, self.p_drop), self._site(gi, SITE_PATH_ATTN, self.dpr[gi]))
                gy = tDa
            self._linear_bwd(f"{k}.proj", rec["att"], Cc, f"{k}.attn.proj.weight", f"{k}.attn.proj.bias", gy, Cc, R, tC,
                             side_tag=f"{k}.proj" if sd else None)
            ad = self._site(gi, SITE_ATTN, self.p_attn)
            if S and prev:                                        # tq / tkv are rewritten by the attention backward below
                self.bwd.add_wait_side(f"{prev}.q")
                self.bwd.add_wait_side(f"{prev}.kv")
            self.bwd.add("ksmi_sr_attention_backward_drop", lambda q=rec["q"], kv=rec["kv"], att=rec["att"], scale=rec["scale"], ad=ad: (
                q.data_ptr(), kv.data_ptr(), att.data_ptr(), tC.data_ptr(), tq.data_ptr(), tkv.data_ptr(), self.scr("attn"), B2, Hs * Ws, 49, heads, Cc, scale,
                ad[0], ad[1], ad[2], self.rng_ptr, dt),
                {"kind": "sr_attention_bwd", "bytes": (6 * R * Cc + 4 * Rk * Cc) * self._es(), "flops": 5 * rec["aflops"] // 2})
            dh = dh_s if S else rec["h2"]                         # single stream: h2 is dead here (fc1 wgrad done), reuse as d(norm1 output)
            if sr > 1:
                Ksr = rec["Ksr"]
                self._linear_bwd(f"{k}.kv", rec["xn"], Cc, f"{k}.attn.kv.weight", f"{k}.attn.kv.bias", tkv, 2 * Cc, Rk, tk,
                                 side_tag=f"{k}.kv" if S else None)
                dxr = dxr_s if S else rec["xn"]                   # single stream: xn is dead after the kv weight gradient
                if S and prev:
                    self.bwd.add_wait_side(f"{prev}.sr")
                self._ln_bwd(tk, rec["xr"], rec["st_sr"], f"{k}.attn.norm.weight", f"{k}.attn.norm.bias", dxr, 0, Rk, Cc)
                dcol2 = self.buf(Rk, Ksr)
                if "sr_wtc" in rec:
                    self._linear_tc_bwd(f"{k}.sr", rec["col2"], Ksr, f"{k}.attn.sr.weight", f"{k}.attn.sr.bias", dxr, Cc, Rk, dcol2, rec["sr_wtc"], Cc, sr * sr,
                                        side_tag=f"{k}.sr" if S else None)
                    self.bwd.add("ksmi_col2im_tc", lambda dcol2=dcol2, dh=dh, Ksr=Ksr: (dcol2.data_ptr(), dh.data_ptr(), 0, B2, Cc, Hs, Ws, Hs // sr, Ws // sr,
                                                                                         sr, sr, sr, 0, Ksr, dt), self._elt_meta("col2im", 2 * Rk * Ksr))
                else:
                    self._linear_bwd(f"{k}.sr", rec["col2"], Ksr, f"{k}.attn.sr.weight", f"{k}.attn.sr.bias", dxr, Cc, Rk, dcol2,
                                     side_tag=f"{k}.sr" if S else None)
                    self.bwd.add("ksmi_col2im", lambda dcol2=dcol2, dh=dh, Ksr=Ksr: (dcol2.data_ptr(), dh.data_ptr(), 0, B2, Cc, Hs, Ws, Hs // sr, Ws // sr,
                                                                                      sr, sr, sr, 0, Ksr, dt), self._elt_meta("col2im", 2 * Rk * Ksr))
                self._linear_bwd(f"{k}.q", rec["h"], Cc, f"{k}.attn.q.weight", f"{k}.attn.q.bias", tq, Cc, R, dh, dx_acc=1,
                                 side_tag=f"{k}.q" if S else None)
            else:
                self._linear_bwd(f"{k}.kv", rec["xn"], Cc, f"{k}.attn.kv.weight", f"{k}.attn.kv.bias", tkv, 2 * Cc, Rk, dh,
                                 side_tag=f"{k}.kv" if S else None)
                self._linear_bwd(f"{k}.q", rec["h"], Cc, f"{k}.attn.q.weight", f"{k}.attn.q.bias", tq, Cc, R, dh, dx_acc=1,
                                 side_tag=f"{k}.q" if S else None)
            self._ln_bwd(dh, rec["t_in"], rec["st1"], f"{k}.norm1.weight", f"{k}.norm1.bias", gt, 1, R, Cc)
            prev = k
        # patch embedding
        self._stage("patch_embed")
        pe = ft["pe"]
        dt0 = tC
        self._ln_bwd(gt, ft["t0"], ft["st_pe"], f"{pe}.norm.weight", f"{pe}.norm.bias", dt0, 0, R, Cc)
        if st == 0:
            self._linear_bwd(f"pe{st + 1}", ft["col"], ft["Kpad"], f"{pe}.proj.weight", f"{pe}.proj.bias", dt0, Cc, R, None, k_real=ft["Kreal"])
        else:
            dcol = ft["col"] if False else self.buf(R, ft["Kpad"])
            prev = self.feats[st - 1]
            if ft.get("pe_wtc") is not None:
                self._linear_tc_bwd(f"pe{st + 1}", ft["col"], ft["Kpad"], f"{pe}.proj.weight", f"{pe}.proj.bias", dt0, Cc, R, dcol, ft["pe_wtc"], prev["C"], 49)
                self.bwd.add("ksmi_col2im_tc", lambda dcol=dcol, prev=prev: (dcol.data_ptr(), prev["df"].data_ptr(), 1, B2, prev["C"], prev["H"], prev["W"],
                                                                             Hs, Ws, 7, 7, 2, 3, ft["Kpad"], dt), self._elt_meta("col2im", 2 * R * ft["Kpad"]))
            else:
                self._linear_bwd(f"pe{st + 1}", ft["col"], ft["Kpad"], f"{pe}.proj.weight", f"{pe}.proj.bias", dt0, Cc, R, dcol)
                self.bwd.add("ksmi_col2im", lambda dcol=dcol, prev=prev: (dcol.data_ptr(), prev["df"].data_ptr(), 1, B2, prev["C"], prev["H"], prev["W"],
                                                                          Hs, Ws, 7, 7, 2, 3, ft["Kpad"], dt), self._elt_meta("col2im", 2 * R * ft["Kpad"]))

    # ---------------------------------------------------------------- convolutions as GEMMs over a channel-fastest im2col
    def _tc_ok(self, Cin):
        """bf16 performance mode: im2col matrices in (tap, channel) order (ksmi_im2col_tc), weights re-ordered per step."""
        return self.dtype == torch.bfloat16 and self.wb is not None and Cin % 8 == 0 and not os.environ.get("KSMI_IM2COL_CT")

    def _linear_tc(self, name, col, Kpad, wkey, bkey, out, N, rows, Cin, taps):
        wtc = torch.empty((N, Kpad), dtype=self.dtype, device=self.dev)
        self.keep.append(wtc)
        w = self.m._p(wkey).data_ptr()
        self.packs.add("ksmi_weight_to_tc", lambda: (w, wtc.data_ptr(), N, Cin, taps, Kpad, self.dt), {"kind": "weight_to_tc", "bytes": 6 * N * Kpad, "flops": 0})
        bp = self.m._p(bkey).data_ptr() if bkey else None
        meta = {"kind": "gemm_nt", "bytes": (rows * Kpad + rows * N + N * Kpad) * 2, "flops": 2 * rows * N * Kpad, "tag": f"{name} K={Kpad} N={N} M={rows}"}
        self.fwd.add("ksmi_gemm_nt", lambda: (col.data_ptr(), Kpad, wtc.data_ptr(), Kpad, bp, None, N, out.data_ptr(), N, rows, Kpad, N), meta)
        return wtc

    def _linear_tc_bwd(self, name, col, Kpad, wkey, bkey, dy, N, rows, dcol, wtc, Cin, taps, side_tag=None):
        if dcol is not None:
            meta = {"kind": "gemm_nn", "bytes": (rows * N + rows * Kpad + N * Kpad) * 2, "flops": 2 * rows * N * Kpad, "tag": f"{name} K={N} N={Kpad} M={rows}"}
            self.bwd.add("ksmi_gemm_nn", lambda: (dy.data_ptr(), N, wtc.data_ptr(), Kpad, dcol.data_ptr(), Kpad, rows, Kpad, N, 0), meta)
        gtc = torch.empty((N, Kpad), dtype=torch.float32, device=self.dev)
        self.keep.append(gtc)
        dw, ws = make_wgrad([SrcSpec(col, Kpad)], dy, N, 0, N, gtc, 1, Kpad, 0, 0, 1, rows, 1, rows, 1, 1, 1, 1, 0, self.dtype)
        self._wgrad(dw, ws, wkey, None if side_tag is None else side_tag + ".tc")
        acc = self._acc_param(wkey)
        g = self.m._g(wkey).data_ptr()
        meta = {"kind": "grad_from_tc", "bytes": 8 * N * Kpad, "flops": 0}
        if side_tag is not None:                     # behind its GEMM on the side stream; the tag marks the pair's end
            meta |= {"side": True, "side_tag": side_tag}
        self.bwd.add("ksmi_grad_from_tc", lambda: (gtc.data_ptr(), g, N, Cin, taps, Kpad, acc), meta)
        self._mark(wkey)
        if bkey:
            self._bias_grad(dy, rows, N, bkey)

    # ---------------------------------------------------------------- execution
    def run_forward(self, x1, x2):
        if x1.data_ptr() != self.xA.data_ptr():
            self.xA.copy_(x1)
        if x2.data_ptr() != self.xB.data_ptr():
            self.xB.copy_(x2)
        self.packs.run()
        self.fwd.run()
        return self.logits

    def run_backward(self, dlogits=None):
        if not self.with_backward:
            raise _lib.KsmiError("plan was built without backward")
        if dlogits is not None and dlogits.data_ptr() != self.dlogits.data_ptr():
            self.dlogits.copy_(dlogits)
        self.bwd.run()
