"""Static launch plan of one MAE forward/backward at a fixed (B, dtype)  (SURVEY.md §8(f) N3).

Reference computation: /root/reference/models/mae.py:54-124.  Token activations are [rows][C] in the activation dtype; the
per-sample random permutation `rand_indices` (int64 [B][N]) is an input of the plan: its first `num_masked` columns are the masked
patch indices, the rest the visible ones (:73-78), and the gather / scatter kernels read those two column ranges in place.

  P0 = patchify(img)                                   to_patch                       (:59)
  E1 = LN(Linear(LN(P0)))                              patch_to_emb                   (:64)
  X0[b][j] = E1[b][vis[b][j]] + pos[1 + vis[b][j]]     + pos_embedding[:, 1:], gather (:65-66, :78)
  Xe = LN_f(encoder layers(X0))                        encoder.transformer            (:86)
  T  = Linear(Xe)                                      enc_to_dec                     (:90)
  DEC0[b][vis] = T + dpos[vis];  DEC0[b][msk] = mask_token + dpos[msk]                (:94-110)
  DEC1 = LN_f(decoder layers(DEC0))                    decoder                        (:111)
  PRED = Linear(DEC1[b][msk])                          to_pixels                      (:115-116)
  loss = mse(PRED, P0[b][msk])                                                        (:82, :122)
"""
import os

import torch

from . import _lib
from .plan_base import PlanBase


class MAEPlan(PlanBase):
    side_tokens = os.environ.get("KSMI_SIDE_TOKENS", "1") != "0"      # plan_base.PlanBase.side_tokens (encoder and decoder layers)
    input_names = ("img", "rand_indices")

    def __init__(self, model, B, dtype, with_backward):
        self._init_base(model, dtype, with_backward)
        self.B = B
        hp, dhp = model.hp, model.dhp
        self.hp, self.dhp = hp, dhp
        self.Cin = hp["channels"]
        self.ih, self.iw = hp["image_size"]
        self.P = hp["patch_size"][0]
        self.N = model.num_patches
        self.nm = model.num_masked
        self.nv = self.N - self.nm
        self.x = torch.empty((B, self.Cin, self.ih, self.iw), dtype=torch.float32, device=self.dev)
        self.idx = torch.zeros((B, self.N), dtype=torch.int64, device=self.dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.dloss = torch.ones(1, dtype=torch.float32, device=self.dev)
        self.mse_ws = torch.zeros(self.lib.ksmi_mse_workspace() // 4, dtype=torch.float32, device=self.dev)
        self._build()
        self._finish()

    def _build(self):
        m, B, dt = self.m, self.B, self.dt
        hp, dhp = self.hp, self.dhp
        D, Dd, N, nm, nv = hp["dim"], dhp["dim"], self.N, self.nm, self.nv
        pd = self.Cin * self.P * self.P
        Rp, Rv, Rm = B * N, B * nv, B * nm
        idx = self.idx
        msk_ptr = lambda: idx.data_ptr()                       # columns [0, nm)
        vis_ptr = lambda: idx.data_ptr() + 8 * nm              # columns [nm, N)
        bwd_steps = []

        # ---- patch embedding (shared with FloodViT: vision_transformer.py:121-126) -----------------------------
        P0, P1, E0, E1 = self.buf(Rp, pd), self.buf(Rp, pd), self.buf(Rp, D), self.buf(Rp, D)
        self.fwd.add("ksmi_patchify", lambda: (self.x.data_ptr(), P0.data_ptr(), B, self.Cin, self.ih, self.iw, self.P, dt),
                     self._elt_meta("patchify", 3 * Rp * pd))
        k = "encoder.to_patch_embedding"
        st_p1 = self._ln(P0, f"{k}.1.weight", f"{k}.1.bias", P1, Rp, pd)
        self._linear("patch_embed", P1, pd, f"{k}.2.weight", f"{k}.2.bias", E0, D, Rp)
        st_p3 = self._ln(E0, f"{k}.3.weight", f"{k}.3.bias", E1, Rp, D)
        # ---- visible tokens + their position rows (mae.py:65, :78) ----------------------------------------------
        X0 = self.buf(Rv, D)
        pos = m._p("encoder.pos_embedding").data_ptr()
        self.fwd.add("ksmi_gather_rows", lambda: (E1.data_ptr(), vis_ptr(), N, X0.data_ptr(), pos, 1, B, N, nv, D, dt),
                     self._elt_meta("gather_rows", 2 * Rv * D))
        self.named.update(patches=P0, embed=E1, x0=X0)
        gxe = self.buf(Rv, D) if self.with_backward else None

        def embed_bwd():
            dE1, dE0, dP1 = self.buf(Rp, D), self.buf(Rp, D), self.buf(Rp, pd)     # dE1 stays zero at the masked positions
            self.bwd.add("ksmi_fill_zero", lambda: (dE1.data_ptr(), dE1.numel() * dE1.element_size()))   # the visible set changes per step
            self.bwd.add("ksmi_scatter_rows", lambda: (gxe.data_ptr(), None, vis_ptr(), N, None, dE1.data_ptr(), B, nv, N, D, dt),
                         self._elt_meta("scatter_rows", 2 * Rv * D))
            # d pos_embedding[0, 1 + t] = sum_b dE1[b][t]  (masked positions contribute zeros); row 0 (cls) gets no gradient
            gpos = m._g("encoder.pos_embedding").data_ptr() + 4 * D
            self.bwd.add("ksmi_batch_sum", lambda: (dE1.data_ptr(), gpos, B, N * D, 0, dt), self._elt_meta("batch_sum", Rp * D))
            self._pinit.add("encoder.pos_embedding")
            self._mark("encoder.pos_embedding")
            self._ln_bwd(dE1, E0, st_p3, f"{k}.3.weight", f"{k}.3.bias", dE0, 0, Rp, D)
            self._linear_bwd("patch_embed", P1, pd, f"{k}.2.weight", f"{k}.2.bias", dE0, D, Rp, dP1)
            self._ln_bwd(dP1, P0, st_p1, f"{k}.1.weight", f"{k}.1.bias", P1, 0, Rp, pd)   # dx of the raw patches is unused
        bwd_steps.append(embed_bwd)

        # ---- encoder transformer + final norm (mae.py:86; vision_transformer.py:84-89) --------------------------
        Xe_in = self._transformer_layers(X0, hp["depth"], "encoder.transformer", B, nv, D, hp["heads"], hp["dim_head"], hp["mlp_dim"],
                                         gxe, bwd_steps, tag="E")
        Xe = self.buf(Rv, D)
        st_e = self._ln(Xe_in, "encoder.transformer.norm.weight", "encoder.transformer.norm.bias", Xe, Rv, D)
        tDe = self.buf(Rv, D) if self.with_backward else None

        # ---- enc_to_dec (mae.py:90) -----------------------------------------------------------------------------
        if D != Dd:
            T = self.buf(Rv, Dd)
            self._linear("enc_to_dec", Xe, D, "enc_to_dec.weight", "enc_to_dec.bias", T, Dd, Rv)
        else:
            T = Xe
        dT = self.buf(Rv, Dd) if self.with_backward else None

        def enc_out_bwd():
            if D != Dd:
                self._linear_bwd("enc_to_dec", Xe, D, "enc_to_dec.weight", "enc_to_dec.bias", dT, Dd, Rv, tDe)
                self._ln_bwd(tDe, Xe_in, st_e, "encoder.transformer.norm.weight", "encoder.transformer.norm.bias", gxe, 0, Rv, D)
            else:
                self._ln_bwd(dT, Xe_in, st_e, "encoder.transformer.norm.weight", "encoder.transformer.norm.bias", gxe, 0, Rv, D)
        bwd_steps.append(enc_out_bwd)

        # ---- decoder tokens (mae.py:94-110) ---------------------------------------------------------------------
        DEC0 = self.buf(Rp, Dd)
        dpos = m._p("decoder_pos_emb.weight").data_ptr()
        mtok = m._p("mask_token").data_ptr()
        self.fwd.add("ksmi_scatter_rows", lambda: (T.data_ptr(), None, vis_ptr(), N, dpos, DEC0.data_ptr(), B, nv, N, Dd, dt),
                     self._elt_meta("scatter_rows", 2 * Rv * Dd))
        self.fwd.add("ksmi_scatter_rows", lambda: (None, mtok, msk_ptr(), N, dpos, DEC0.data_ptr(), B, nm, N, Dd, dt),
                     self._elt_meta("scatter_rows", Rm * Dd))
        gxd = self.buf(Rp, Dd) if self.with_backward else None
        dMT = self.buf(Rm, Dd) if self.with_backward else None

        def dec_in_bwd():
            # gxd = d DEC0: visible rows -> dT, masked rows -> d mask_token, every row -> d decoder_pos_emb
            self.bwd.add("ksmi_gather_rows", lambda: (gxd.data_ptr(), vis_ptr(), N, dT.data_ptr(), None, 0, B, N, nv, Dd, dt),
                         self._elt_meta("gather_rows", 2 * Rv * Dd))
            self.bwd.add("ksmi_gather_rows", lambda: (gxd.data_ptr(), msk_ptr(), N, dMT.data_ptr(), None, 0, B, N, nm, Dd, dt),
                         self._elt_meta("gather_rows", 2 * Rm * Dd))
            a1 = self._acc_param("mask_token")
            gm = m._g("mask_token").data_ptr()
            self.bwd.add("ksmi_colsum", lambda: (dMT.data_ptr(), Rm, Dd, gm, a1, dt), self._elt_meta("colsum", Rm * Dd))
            a2 = self._acc_param("decoder_pos_emb.weight")
            gp = m._g("decoder_pos_emb.weight").data_ptr()
            self.bwd.add("ksmi_batch_sum", lambda: (gxd.data_ptr(), gp, B, N * Dd, a2, dt), self._elt_meta("batch_sum", Rp * Dd))
            self._mark("mask_token", "decoder_pos_emb.weight")
        bwd_steps.append(dec_in_bwd)

        # ---- decoder transformer + final norm (mae.py:111) -------------------------------------------------------
        Xd_in = self._transformer_layers(DEC0, dhp["depth"], "decoder", B, N, Dd, dhp["heads"], dhp["dim_head"], dhp["mlp_dim"], gxd,
                                         bwd_steps, tag="D")
        DEC1 = self.buf(Rp, Dd)
        st_d = self._ln(Xd_in, "decoder.norm.weight", "decoder.norm.bias", DEC1, Rp, Dd)

        # ---- masked tokens -> pixel values, loss (mae.py:113-122) ------------------------------------------------
        MT, PRED, TGT = self.buf(Rm, Dd), self.buf(Rm, pd), self.buf(Rm, pd)
        self.fwd.add("ksmi_gather_rows", lambda: (DEC1.data_ptr(), msk_ptr(), N, MT.data_ptr(), None, 0, B, N, nm, Dd, dt),
                     self._elt_meta("gather_rows", 2 * Rm * Dd))
        self._linear("to_pixels", MT, Dd, "to_pixels.weight", "to_pixels.bias", PRED, pd, Rm)
        self.fwd.add("ksmi_gather_rows", lambda: (P0.data_ptr(), msk_ptr(), N, TGT.data_ptr(), None, 0, B, N, nm, pd, dt),
                     self._elt_meta("gather_rows", 2 * Rm * pd))
        self.fwd.add("ksmi_mse_loss", lambda: (PRED.data_ptr(), TGT.data_ptr(), None, 1.0, None, self.loss.data_ptr(), self.mse_ws.data_ptr(),
                                               Rm * pd, dt), self._elt_meta("mse", 2 * Rm * pd))
        self.named.update(dec0=DEC0, dec1=DEC1, pred=PRED, target=TGT, enc_out=Xe)
        if not self.with_backward:
            return

        # ---- backward ---------------------------------------------------------------------------------------------
        dPRED, dMTo, dDEC1 = self.buf(Rm, pd), self.buf(Rm, Dd), self.buf(Rp, Dd)      # dDEC1 stays zero at the visible positions
        scratch_loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.keep.append(scratch_loss)
        self.bwd.add("ksmi_mse_loss", lambda: (PRED.data_ptr(), TGT.data_ptr(), dPRED.data_ptr(), 1.0, self.dloss.data_ptr(),
                                               scratch_loss.data_ptr(), self.mse_ws.data_ptr(), Rm * pd, dt), self._elt_meta("mse_bwd", 3 * Rm * pd))
        self._linear_bwd("to_pixels", MT, Dd, "to_pixels.weight", "to_pixels.bias", dPRED, pd, Rm, dMTo)
        self.bwd.add("ksmi_fill_zero", lambda: (dDEC1.data_ptr(), dDEC1.numel() * dDEC1.element_size()))   # the masked set changes per step
        self.bwd.add("ksmi_scatter_rows", lambda: (dMTo.data_ptr(), None, msk_ptr(), N, None, dDEC1.data_ptr(), B, nm, N, Dd, dt),
                     self._elt_meta("scatter_rows", 2 * Rm * Dd))
        self._ln_bwd(dDEC1, Xd_in, st_d, "decoder.norm.weight", "decoder.norm.bias", gxd, 0, Rp, Dd)
        for step in reversed(bwd_steps):
            step()

    # ---------------------------------------------------------------- execution
    def run_forward(self, img, rand_indices):
        if img.data_ptr() != self.x.data_ptr():
            self.x.copy_(img)
        if rand_indices.data_ptr() != self.idx.data_ptr():
            self.idx.copy_(rand_indices)
        self.packs.run()
        self.fwd.run()
        return self.loss

    def run_backward(self, dloss=None):
        if not self.with_backward:
            raise _lib.KsmiError("plan was built without backward")
        if dloss is not None:
            self.dloss.copy_(dloss.reshape(1))
        self.bwd.run()
