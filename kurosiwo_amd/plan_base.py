"""Shared machinery of the static launch plans (FloodViT, ChangeFormer): scratch bookkeeping, weight packing, the
launch-list wrappers around ksmi_conv_forward / ksmi_conv_wgrad for nn.Linear, nn.LayerNorm and ConvTranspose2d(k4,s2,p1)."""
import ctypes as C
import os

import torch

from . import _lib
from .runtime import DT, SrcSpec, conv_npad, make_conv, make_pack, make_wgrad, packed_weight_numel
from .snunet_plan import LaunchList

LN_EPS = 1e-5


class PlanBase:
    # Weight-gradient launches may run on the train step's side stream (trainer.py overlap_wgrad, snunet_plan.StepStreams) only in plans where (a) every user of
    # the "wgrad" scratch goes through _wgrad (true here: the lane keeps them in order) and (b) no operand of a weight gradient is
    # rewritten by a later launch of the same backward pass.  (b) holds for the convolutional plans, whose activations and gradients
    # are dedicated buffers; the token plans (ChangeFormer encoder, FloodViT, MAE) recycle their per-block gradient buffers.
    side_wgrad = False
    # The token plans instead name the weight gradients that may leave the critical path one by one (side_tokens; _linear_bwd side_tag)
    # and place a wait (LaunchList.add_wait_side) before every launch that overwrites an operand of one of them.
    side_tokens = False
    slab_bias_side = False     # bias rows of split-mode token weight gradients also from side-stream launches (_linear_wgrad)

    def _init_base(self, model, dtype, with_backward):
        self.m, self.dtype, self.with_backward = model, dtype, with_backward
        self.dev = model.flat_params.device
        self.dt = DT[dtype]
        self.lib = _lib.load()
        self.packs, self.fwd, self.bwd = LaunchList(), LaunchList(), LaunchList()
        self.keep, self._pack_descs, self._pinit = [], [], set()
        self.param_ready = {}
        self.named = {}            # debug/test access to intermediate activations
        self._need, self._bufs, self._later = {}, {}, []
        # deferred row reductions (LayerNorm dgamma / dbeta, long-axis bias and depthwise weight gradients): one batched launch per
        # `rowsum_batch` producer launches instead of one tiny launch each (single-stream plans pay every launch in full)
        self._rs_entries, self._rs_slots = [], 0
        # (read per plan, not at import: tests and A/B runs vary them between plans of one process)
        self.csum_rows = max(1, int(os.environ.get("KSMI_CSUM_ROWS", "512")))
        self.rowsum_batch = max(1, int(os.environ.get("KSMI_ROWSUM_BATCH", "16")))
        # bf16 mirror of the parameter arena for the token GEMMs (gemm.hip): one cast launch per step
        self.wb = None
        if dtype == torch.bfloat16 and not os.environ.get("KSMI_LINEAR_IGEMM"):
            n = model.flat_params.numel()
            assert n % 8 == 0
            self.wb = torch.empty(n, dtype=torch.bfloat16, device=self.dev)
            fp, wb = model.flat_params, self.wb
            # The cast is skipped for the forward that follows an optimiser step which wrote the mirror itself (ksmi_adam_step_mirror:
            # trainer -> mirror_written()).  Freshness = the version counter of the fp32 arena at that moment: any in-place torch
            # operation on a parameter afterwards (load_state_dict, init, .copy_) bumps it and the cast runs again; the mark is
            # consumed by one forward, so an evaluation pass or a second forward re-casts too.  KSMI_ADAM_MIRROR=0: always cast.
            self._mirror_version = None
            self.packs.add("ksmi_cast_bf16", lambda: (fp.data_ptr(), wb.data_ptr(), n),
                           {"kind": "cast_bf16", "bytes": 6 * n, "flops": 0, "skip_if": self._mirror_is_fresh})

    def mirror_ptr(self):
        """device pointer of the bf16 operand copy of the parameter arena for an optimiser that writes it (None: this plan has none)"""
        if self.wb is None or os.environ.get("KSMI_ADAM_MIRROR", "1") == "0" or getattr(self, "_mirror_off", False):
            return None
        return self.wb.data_ptr()

    def mirror_written(self):
        """the optimiser step that just ran wrote THIS plan's operand copy: record the arena's generation (optim.arena_generation: bumped by
        every fused optimiser step on the arena, whichever plan drove it) and torch's version counter (bumped by in-place torch edits)"""
        from .optim import arena_generation
        fp = self.m.flat_params
        self._mirror_version = (arena_generation(fp.data_ptr()), fp._version)

    def _mirror_is_fresh(self):
        """True once: the last write of the parameters was this plan's own mirrored step.  A step of ANOTHER plan of the same model (a
        different batch shape: trainers rebuild their step when B changes, the plans stay cached in model._plans) bumps the generation
        and writes only ITS copy, so this plan casts again; so does any in-place torch edit of a parameter."""
        from .optim import arena_generation
        v, self._mirror_version = getattr(self, "_mirror_version", None), None
        fp = self.m.flat_params
        return v is not None and v == (arena_generation(fp.data_ptr()), fp._version)

    def _wb_ptr(self, key):
        return self.wb.data_ptr() + 2 * self.m._poff[key]

    csum_rows, rowsum_batch = 512, 16          # class defaults; _init_base reads KSMI_CSUM_ROWS / KSMI_ROWSUM_BATCH per plan

    def _finish(self):
        self._flush_rowsums()
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
    def _stage(self, name):
        """measurement tag of the launches appended from here on (bench.py roofline.stages / attention_block)"""
        self.fwd.cur_stage = self.bwd.cur_stage = name

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

    def _acc_param(self, key, deferred=False):
        if not deferred and any(e["key"] == key for e in self._rs_entries):
            self._flush_rowsums()                          # a direct writer behind a deferred one: the deferred sum lands first
        acc = 1 if key in self._pinit else 0
        self._pinit.add(key)
        return acc

    def _mark(self, *keys):
        for k in keys:
            self.param_ready[k] = len(self.bwd.pending) - 1

    def _rs_slot(self, nbytes):
        """scratch of the next deferred row reduction's partial rows; recycled after the batched launch that consumes it"""
        name = f"rs{self._rs_slots}"
        self._rs_slots += 1
        self.need(name, nbytes)
        return name

    def _defer_rowsum(self, key, slot, off, rows, K, k, Cstride, Cc, side_ok=False):
        """grad[key][c] (+)= sum_r partial[(r*K + k)*Cstride + c], partial = scratch `slot` + off bytes (written by the launch just
        appended to self.bwd), in the next batched reduction.  Entries of one key inside a batch chain behind the first."""
        # the recycled slots and the batched reducer rely on list order on ONE stream: a producer handed to the side stream would race
        # the slot's next owner and the reducer
        # -- unless the caller says so (side_ok: the split-mode token weight gradient that also writes the bias rows): the batched reducer
        # of this batch then waits for the whole side stream first, and the slot's next owner is issued behind that reducer either way
        side = bool(self.bwd.pending and self.bwd.pending[-1][2].get("side"))
        if side and not side_ok:
            raise AssertionError(f"deferred row sum of {key}: its producer launch is side-stream tagged")
        acc = self._acc_param(key, deferred=True)
        prev = [i for i, e in enumerate(self._rs_entries) if e["key"] == key]
        self._rs_entries.append(dict(key=key, slot=slot, off=off, rows=rows, K=K, k=k, Cstride=Cstride, C=Cc,
                                     accumulate=0 if prev else acc, head=0 if prev else 1, next=-1, side=side))
        if prev:
            self._rs_entries[prev[-1]]["next"] = len(self._rs_entries) - 1

    def _rs_tick(self):
        if self._rs_slots >= self.rowsum_batch:
            self._flush_rowsums()

    def _flush_rowsums(self):
        ents, self._rs_entries, self._rs_slots = self._rs_entries, [], 0
        if not ents:
            return
        n, max_c = len(ents), max(e["C"] for e in ents)
        holder = {}

        def build():
            arr = (_lib.RowsumDesc * n)()
            for r, e in zip(arr, ents):
                r.partial, r.dst = self.scr(e["slot"]) + e["off"], self.m._g(e["key"]).data_ptr()
                r.rows, r.K, r.k, r.Cstride, r.C = e["rows"], e["K"], e["k"], e["Cstride"], e["C"]
                r.accumulate, r.head, r.next = e["accumulate"], e["head"], e["next"]
            raw = bytes(C.string_at(C.addressof(arr), C.sizeof(arr)))
            holder["t"] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
            self.keep.append(holder["t"])
        self._later.append(build)
        nbytes = sum(e["rows"] * e["C"] * 4 for e in ents)
        if any(e.get("side") for e in ents):
            self.bwd.add_wait_side(None)         # partial rows written on the side stream (split-mode weight gradients with bias rows)
        self.bwd.add("ksmi_reduce_rows_batched_wide", lambda: (holder["t"].data_ptr(), n, max_c),
                     {"kind": "reduce_rows", "bytes": nbytes, "flops": 0})
        self._mark(*[e["key"] for e in ents])

    def _packed(self, key, table, taps, N, n_mod, sK, sN, sD, sT, flip=0, tap_map=None):
        Npad = conv_npad(N)
        out = torch.empty(packed_weight_numel(table, taps, Npad, self.dtype), dtype=self.dtype, device=self.dev)
        d = make_pack(self.m._p(key), out, table, taps, N, Npad, n_mod, sK, sN, sD, sT, flip, tap_map)
        self.keep += [d, out]
        self._pack_descs.append(d)
        return out

    def _conv(self, ll, d, tag, name=""):
        self.keep.append(d)
        d.dir = 1 if "dgrad" in tag else 0            # (profiling tag: kernel names carry the direction, ksmi.h)
        taps, es = d.KH * d.KW, self._es()
        ktot = sum(d.src[i].c_len for i in range(d.nsrc))
        pin, pout = d.B * d.Hin * d.Win, d.B * d.Hout * d.Wout
        elems = pin * ktot + sum(pout * d.dst[i].n_len * (2 if d.dst[i].accumulate else 1) for i in range(d.ndst))
        if d.mask_src:
            elems += pout * d.N
        meta = {"kind": f"igemm_{tag}<{d.KH}x{d.KW}s{d.stride}>", "bytes": elems * es + taps * ktot * d.N * es,
                "flops": 2 * pout * d.N * ktot * taps, "tag": f"{name} K={ktot} N={d.N} M={pout}"}
        ll.add("ksmi_conv_forward", lambda: (C.byref(d), self.dt), meta)

    def _wgrad(self, d, ws, key, side_tag=None):
        if side_tag is None and self.side_tokens:
            self.bwd.add_wait_side(None)         # shares the split-slab scratch with the side-stream gradients
        self.keep.append(d)
        self.need("wgrad", ws)
        self._later.append(lambda: setattr(d, "partial", self.scr("wgrad")))
        taps, es = d.KH * d.KW, self._es()
        ktot = sum(d.src[i].c_len for i in range(d.nsrc))
        pin, pout = d.B * d.Hin * d.Win, d.B * d.Hout * d.Wout
        meta = {"kind": f"igemm_wgrad<{d.KH}x{d.KW}s{d.stride}>", "bytes": (pin * ktot + pout * d.N) * es + taps * ktot * d.N * 4,
                "flops": 2 * pout * d.N * ktot * taps, "tag": f"{key} K={ktot} N={d.N} M={pout}"}
        if self.side_wgrad or side_tag is not None:
            meta["side"] = True              # snunet_plan.LaunchList.run: eligible for the side stream
        if side_tag is not None:
            meta["side_tag"] = side_tag
        self.bwd.add("ksmi_conv_wgrad", lambda: (C.byref(d), self.dt), meta)
        self._mark(key)

    def _elt_meta(self, kind, nelem_rw):
        return {"kind": kind, "bytes": int(nelem_rw) * self._es(), "flops": 0}

    # ---------------------------------------------------------------- nn.Linear on token rows
    def _linear(self, name, x, Cin, wkey, bkey, out, N, rows, resid=None, k_real=None):
        """out[rows, N] = x[rows, Cin] @ W[N, k_real]^T + b [+ resid]; columns of x beyond k_real (default Cin) are zero padding."""
        kr = Cin if k_real is None else k_real
        if self.wb is not None and kr == Cin and Cin % 8 == 0 and N % 8 == 0:
            wp = self._wb_ptr(wkey)
            bp = self.m._p(bkey).data_ptr() if bkey else None
            es = 2
            meta = {"kind": "gemm_nt", "bytes": (rows * Cin + rows * N * (2 if resid is not None else 1) + N * Cin) * es,
                    "flops": 2 * rows * N * Cin, "tag": f"{name} K={Cin} N={N} M={rows}"}
            self.fwd.add("ksmi_gemm_nt", lambda: (x.data_ptr(), Cin, wp, Cin, bp, None if resid is None else resid.data_ptr(), N,
                                                  out.data_ptr(), N, rows, Cin, N), meta)
            return
        d, table = make_conv([SrcSpec(x, Cin, k_real=kr)], [(out, N, 0, 0, N, 0)], out, self.m._p(bkey) if bkey else None, None,
                             1, rows, 1, rows, 1, 1, 1, 1, 0, N, self.dtype, resid=None if resid is None else (resid, N))
        d.wpk = self._packed(wkey, table, 1, N, N, 1, kr, 0, 0).data_ptr()
        self._conv(self.fwd, d, "linear", name)

    def _bias_grad(self, dy, rows, N, bkey):
        if rows <= 8192:                                   # token matrices: one launch (long pixel axes: two-stage below)
            acc = self._acc_param(bkey)
            gb = self.m._g(bkey).data_ptr()
            self.bwd.add("ksmi_colsum", lambda: (dy.data_ptr(), rows, N, gb, acc, self.dt), self._elt_meta("colsum", rows * N))
            self._mark(bkey)
            return
        r = max(1, min(self.csum_rows, rows // 64))        # partial rows = workgroups of the streaming kernel (the batched reducer takes any count)
        slot = self._rs_slot(r * N * 4)
        self.bwd.add("ksmi_channel_sum", lambda: (dy.data_ptr(), self.scr(slot), r, rows, N, self.dt),
                     self._elt_meta("channel_sum", rows * N))
        self._defer_rowsum(bkey, slot, 0, r, 1, 0, N, N)
        self._rs_tick()

    def _linear_bwd(self, name, x, Cin, wkey, bkey, dy, N, rows, dx, want_w=True, dx_acc=0, k_real=None, side_tag=None, wait_tag=None):
        """dx (+)= dy @ W (skipped if dx is None) ; dW = dy^T x ; db = colsum(dy).
        side_tag (plans with side_tokens): the weight gradient is listed first and may run on the side stream next to its own input
        gradient; the caller owns the hazards (add_wait_side(side_tag) before x or dy is overwritten).  wait_tag: an earlier
        side-stream gradient that reads the buffer this input gradient writes."""
        if not self.side_tokens:
            side_tag = wait_tag = None
        if side_tag is not None and want_w:
            self._linear_wgrad(x, Cin, wkey, bkey, dy, N, rows, k_real, side_tag)
            want_w = False
        if wait_tag is not None:
            self.bwd.add_wait_side(wait_tag)
        kr = Cin if k_real is None else k_real
        if dx is not None and self.wb is not None and kr == Cin and Cin % 8 == 0 and N % 8 == 0:
            wp = self._wb_ptr(wkey)
            meta = {"kind": "gemm_nn", "bytes": (rows * N + rows * Cin * (2 if dx_acc else 1) + N * Cin) * 2,
                    "flops": 2 * rows * N * Cin, "tag": f"{name} K={N} N={Cin} M={rows}"}
            self.bwd.add("ksmi_gemm_nn", lambda: (dy.data_ptr(), N, wp, Cin, dx.data_ptr(), Cin, rows, Cin, N, dx_acc), meta)
        elif dx is not None:
            d, table = make_conv([SrcSpec(dy, N)], [(dx, Cin, 0, 0, Cin, dx_acc)], dx, None, None,
                                 1, rows, 1, rows, 1, 1, 1, 1, 0, Cin, self.dtype)
            d.wpk = self._packed(wkey, table, 1, Cin, Cin, Cin, 1, 0, 0).data_ptr()
            self._conv(self.bwd, d, "linear_dgrad", name)
        if want_w:
            self._linear_wgrad(x, Cin, wkey, bkey, dy, N, rows, k_real, None)

    def _linear_wgrad(self, x, Cin, wkey, bkey, dy, N, rows, k_real, side_tag):
        kr = Cin if k_real is None else k_real
        dw, ws = make_wgrad([SrcSpec(x, Cin, k_real=kr)], dy, N, 0, N, self.m._g(wkey), 1, kr, 0, self._acc_param(wkey),
                            1, rows, 1, rows, 1, 1, 1, 1, 0, self.dtype)
        fused = 0
        if bkey:       # the bias gradient rides on the weight-gradient GEMM when that launch holds dY in LDS anyway (gemm2.hip): 1 = direct
                       # mode writes it; 2 = split mode writes one partial row per split, summed by the next batched row reduction
            dw.bias_grad = self.m._g(bkey).data_ptr()
            fused = int(self.lib.ksmi_conv_wgrad_fuses_bias(C.byref(dw), self.dt))
            # ... from a side-stream producer only in the plans that gain from it (slab_bias_side): the batched reducer then waits for
            # the side stream -- ChangeFormer + 1.1 % (51 channel_sum passes leave the critical stream), FloodViT - 1.4 % (its only split
            # layer is the 2-split proj gradient: 24 waits per step for 24 colsum launches of 10 us).  KSMI_SLAB_BIAS_SIDE=0 / 1 forces.
            if fused == 2 and (side_tag is not None or self.side_wgrad):
                force = os.environ.get("KSMI_SLAB_BIAS_SIDE")
                if not (self.slab_bias_side if force is None else force != "0"):
                    fused = 0
            if fused == 1:
                dw.bias_accumulate = self._acc_param(bkey)
            elif fused == 2:
                slot = self._rs_slot(dw.nsplit * N * 4)
                dw.bias_grad = None
                self._later.append(lambda: setattr(dw, "bias_grad", self.scr(slot)))
            else:
                dw.bias_grad = None
        self._wgrad(dw, ws, wkey, side_tag)
        if bkey and fused == 1:
            self._mark(bkey)
        elif bkey and fused == 2:
            self._defer_rowsum(bkey, slot, 0, dw.nsplit, 1, 0, N, N, side_ok=True)
            self._rs_tick()
        elif bkey:
            self._bias_grad(dy, rows, N, bkey)

    # ---------------------------------------------------------------- nn.LayerNorm
    def _ln(self, x, wkey, bkey, y, rows, Cc, eps=LN_EPS):
        st = self.fbuf(2, rows)
        g, b = self.m._p(wkey).data_ptr(), self.m._p(bkey).data_ptr()
        self.fwd.add("ksmi_layernorm_forward", lambda: (x.data_ptr(), g, b, y.data_ptr(), st[0].data_ptr(), st[1].data_ptr(),
                                                        rows, Cc, eps, self.dt), self._elt_meta("layernorm_fwd", 2 * rows * Cc))
        return st

    def _ln_bwd(self, dy, x, st, wkey, bkey, dx, accumulate, rows, Cc, want_w=True):
        nblk = self.lib.ksmi_layernorm_bwd_blocks(rows)
        if want_w:
            slot = self._rs_slot(nblk * 2 * Cc * 4)
        else:
            slot = "lnp"
            self.need("lnp", nblk * 2 * Cc * 4)
        g = self.m._p(wkey).data_ptr()
        self.bwd.add("ksmi_layernorm_backward", lambda: (dy.data_ptr(), x.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), g,
                                                         dx.data_ptr(), accumulate, self.scr(slot), rows, Cc, self.dt),
                     self._elt_meta("layernorm_bwd", (3 + accumulate) * rows * Cc))
        if want_w:                                       # partial rows [nblk][2][C]: k = 0 -> dbeta, k = 1 -> dgamma
            self._defer_rowsum(bkey, slot, 0, nblk, 2, 0, Cc, Cc)
            self._defer_rowsum(wkey, slot, 0, nblk, 2, 1, Cc, Cc)
            self._rs_tick()

    # ---------------------------------------------------------------- pre-norm transformer layers
    def _transformer_layers(self, X, depth, prefix, B, Ntok, D, heads, dim_head, M, gx, bwd_steps, tag="L"):
        """depth x [x += Attention(LN(x)); x += FeedForward(LN(x))] on token rows [B*Ntok][D] (vision_transformer.py:35-66,19-32,
        84-88; the final Transformer.norm is the caller's).  Parameter names follow `{prefix}.layers.{i}.0|1.*`.  Backward closures
        are appended to `bwd_steps` in forward order (the caller runs them reversed); they update `gx`, the gradient of the
        residual stream, in place."""
        R, I = B * Ntok, heads * dim_head
        dt = self.dt
        if dim_head != 64:
            raise _lib.KsmiError("attention kernel is specialised for dim_head = 64")
        if self.with_backward:
            tD, tI, tM, tQ = self.buf(R, D), self.buf(R, I), self.buf(R, M), self.buf(R, 3 * I)
        scale = float(dim_head) ** -0.5
        for li in range(depth):
            a, f = f"{prefix}.layers.{li}.0", f"{prefix}.layers.{li}.1"
            x_in = X
            h1, qkv, att, x_mid = self.buf(R, D), self.buf(R, 3 * I), self.buf(R, I), self.buf(R, D)
            h2, u, g, x_out = self.buf(R, D), self.buf(R, M), self.buf(R, M), self.buf(R, D)
            lse = self.fbuf(B, heads, Ntok)
            st1 = self._ln(x_in, f"{a}.norm.weight", f"{a}.norm.bias", h1, R, D)
            self._linear(f"{tag}{li}.to_qkv", h1, D, f"{a}.to_qkv.weight", None, qkv, 3 * I, R)
            aflops = 4 * B * heads * Ntok * Ntok * 64
            self.fwd.add("ksmi_attention_forward", lambda qkv=qkv, att=att, lse=lse: (
                qkv.data_ptr(), att.data_ptr(), lse.data_ptr(), B, Ntok, heads, 64, scale, dt),
                {"kind": "attention_fwd", "bytes": 4 * R * I * self._es(), "flops": aflops})
            # the residual adds ride in the GEMM epilogues (one rounding of acc + bias + x instead of two)
            self._linear(f"{tag}{li}.to_out", att, I, f"{a}.to_out.0.weight", f"{a}.to_out.0.bias", x_mid, D, R, resid=x_in)
            st2 = self._ln(x_mid, f"{f}.net.0.weight", f"{f}.net.0.bias", h2, R, D)
            self._linear(f"{tag}{li}.ff1", h2, D, f"{f}.net.1.weight", f"{f}.net.1.bias", u, M, R)
            self.fwd.add("ksmi_gelu_forward", lambda u=u, g=g: (u.data_ptr(), g.data_ptr(), R * M, dt), self._elt_meta("gelu", 2 * R * M))
            self._linear(f"{tag}{li}.ff2", g, M, f"{f}.net.4.weight", f"{f}.net.4.bias", x_out, D, R, resid=x_mid)
            X = x_out
            self.named[f"layer{li}" if tag == "L" else f"{tag}layer{li}"] = x_out

            def layer_bwd(li=li, a=a, f=f, x_in=x_in, h1=h1, qkv=qkv, att=att, x_mid=x_mid, h2=h2, u=u, g=g, lse=lse,
                          st1=st1, st2=st2, aflops=aflops):
                # Side-stream weight gradients (side_tokens): the per-layer activations they read live until the next forward; the
                # gradient buffers gx / tM / tQ are shared by all layers, so the launches that overwrite them wait for the reader:
                #   ff2 (reads gx) before the LayerNorm backward that adds into gx; ff1 (reads tM) before the NEXT layer's ff2 input
                #   gradient writes tM; to_out (reads gx) before the second LayerNorm backward; to_qkv (reads tQ) before the NEXT
                #   layer's attention backward writes tQ.
                nm = lambda k, l=li: f"{tag}{l}.{k}"
                later = li + 1 < depth                    # the layer processed just before this one in the backward pass
                # FeedForward: x_out = x_mid + W2 gelu(W1 LN(x_mid) + b1) + b2
                self._linear_bwd(nm("ff2"), g, M, f"{f}.net.4.weight", f"{f}.net.4.bias", gx, D, R, tM, side_tag=nm("ff2"),
                                 wait_tag=nm("ff1", li + 1) if later else None)
                self.bwd.add("ksmi_gelu_backward", lambda: (tM.data_ptr(), u.data_ptr(), tM.data_ptr(), R * M, dt),
                             self._elt_meta("gelu_bwd", 3 * R * M))
                self._linear_bwd(nm("ff1"), h2, D, f"{f}.net.1.weight", f"{f}.net.1.bias", tM, M, R, tD, side_tag=nm("ff1"))
                if self.side_tokens:
                    self.bwd.add_wait_side(nm("ff2"))
                self._ln_bwd(tD, x_mid, st2, f"{f}.net.0.weight", f"{f}.net.0.bias", gx, 1, R, D)
                # Attention: x_mid = x_in + Wo attn(Wqkv LN(x_in)) + bo
                self._linear_bwd(nm("to_out"), att, I, f"{a}.to_out.0.weight", f"{a}.to_out.0.bias", gx, D, R, tI, side_tag=nm("to_out"))
                self.need("attn", self.lib.ksmi_attention_bwd_workspace(B, Ntok, heads, 64, dt))
                if self.side_tokens and later:
                    self.bwd.add_wait_side(nm("to_qkv", li + 1))
                self.bwd.add("ksmi_attention_backward", lambda: (qkv.data_ptr(), att.data_ptr(), lse.data_ptr(), tI.data_ptr(),
                                                                 tQ.data_ptr(), self.scr("attn"), B, Ntok, heads, 64, scale, dt),
                             {"kind": "attention_bwd", "bytes": 8 * R * I * self._es(), "flops": 5 * aflops // 2})
                self._linear_bwd(nm("to_qkv"), h1, D, f"{a}.to_qkv.weight", None, tQ, 3 * I, R, tD, side_tag=nm("to_qkv"))
                if self.side_tokens:
                    self.bwd.add_wait_side(nm("to_out"))
                self._ln_bwd(tD, x_in, st1, f"{a}.norm.weight", f"{a}.norm.bias", gx, 1, R, D)
            bwd_steps.append(layer_bwd)
        return X

    # ---------------------------------------------------------------- ConvTranspose2d(k4, s2, p1)
    def _deconv(self, name, x, Cin, N, H, W, out, outC, prefix="head.", suffix="", B=None):
        """out[B,2H,2W,outC][..., :N] = ConvTranspose2d(x) + bias as 4 phase convolutions with 2x2 taps:
        out[2m+py] = sum_a x[m - pad + a] * W[ky],  pad = 1 - py,  ky = (3 - 2a) if py == 0 else (2 - 2a)."""
        wkey, bkey = f"{prefix}{name}{suffix}.weight", f"{prefix}{name}{suffix}.bias"
        B = self.B if B is None else B
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

    def _deconv_wgrad(self, src, x, Cin, N, H, W, wkey, B):
        """dW[c][n][ky][kx] = sum x[iy,ix,c] dOut[2iy-1+ky, 2ix-1+kx, n] as four 2x2 stride-1 weight-gradient GEMMs over the parity
        sub-images of dOut (tap (a,b) of phase (py,px) <-> ky = 2a | 1+2a, kx = 2b | 1+2b), each writing its 4 of the 16 taps"""
        acc = self._acc_param(wkey)
        if os.environ.get("KSMI_DECONV_WGRAD_4X4"):
            dw, ws = make_wgrad(src, x, Cin, 0, Cin, self.m._g(wkey), 16, N * 16, 1, acc, B, 2 * H, 2 * W, H, W, 4, 4, 2, 1, self.dtype)
            self._wgrad(dw, ws, wkey)
            return
        for py in range(2):
            for px in range(2):
                tap_off = [(2 * a if py else 1 + 2 * a) * 4 + (2 * b if px else 1 + 2 * b) for a in range(2) for b in range(2)]
                dw, ws = make_wgrad(src, x, Cin, 0, Cin, self.m._g(wkey), 16, N * 16, 0, acc, B, H, W, H, W, 2, 2, 1, py, self.dtype,
                                    pad_x=px, in_map=(2, 2, py, px, 2 * H, 2 * W), tap_off=tap_off)
                self._wgrad(dw, ws, wkey)

    def _deconv_bwd(self, name, x, Cin, N, H, W, dout, doutC, dx, mask=None, prefix="head.", suffix="", B=None, stats=None):
        """dout [B,2H,2W,doutC] (first N channels real).  dx[B,H,W,Cin] = 4x4 stride-2 conv of dout (optionally
        ReLU-masked by `mask`), dW via the stride-2 weight-gradient GEMM, db = channel sums."""
        wkey, bkey = f"{prefix}{name}{suffix}.weight", f"{prefix}{name}{suffix}.bias"
        B = self.B if B is None else B
        src = [SrcSpec(dout, doutC, 0, doutC, k_real=N)]
        if dx is not None:
            mk = None
            if mask is not None:
                mk = (mask, self.const[0], self.const[1], self.const[1], self.const[0])
            # one 4x4 stride-2 convolution for the narrow decoders (FloodViT head, N <= 128 output channels of the deconv: the four
            # phase launches each re-read and re-write dx; measured 160 against 4 x 80 us on the 64 <- 3 layer, level on the others),
            # four dense 2x2 phase convolutions for ChangeFormer's 256-channel decoder.  KSMI_DECONV_DGRAD_4X4=0/1 forces one form.
            force = os.environ.get("KSMI_DECONV_DGRAD_4X4")
            if force == "1" or (force is None and N <= 128):
                d, table = make_conv(src, [(dx, Cin, 0, 0, Cin, 0)], dx, None, None, B, 2 * H, 2 * W, H, W, 4, 4, 2, 1, Cin,
                                     self.dtype, mask=mk)
                d.wpk = self._packed(wkey, table, 16, Cin, Cin, 16, N * 16, 0, 1, 0).data_ptr()
                self._conv(self.bwd, d, "deconv_dgrad", name)
            else:
                # dIn[iy] = sum_ky dOut[2 iy - 1 + ky] W[ky]: split by the parity of the dOut row.  Even rows E[m] = dOut[2m]: taps a = 0,1
                # read E[iy + a] with ky = 1 + 2a (pad 0); odd rows O[m] = dOut[2m+1]: taps read O[iy - 1 + a] with ky = 2a (pad 1).
                # Four dense 2x2 stride-1 convolutions over the strided views (every M-tile row useful; the single 4x4 stride-2
                # convolution is limited to 100-pixel patches by its 22x22 halo = 39 % of the 256-row tile), accumulated in dx.
                first = True
                for py in range(2):
                    for px in range(2):
                        tap_map = []
                        for a in range(2):
                            for b in range(2):
                                ky = 2 * a if py else 1 + 2 * a
                                kx = 2 * b if px else 1 + 2 * b
                                tap_map.append(ky * 4 + kx)
                        d, table = make_conv(src, [(dx, Cin, 0, 0, Cin, 0 if first else 1)], dx, None, None, B, H, W, H, W, 2, 2, 1, py, Cin,
                                             self.dtype, mask=mk, pad_x=px, in_map=(2, 2, py, px, 2 * H, 2 * W))   # 0/1 mask on every partial = mask on the sum
                        d.wpk = self._packed(wkey, table, 4, Cin, Cin, 16, N * 16, 0, 1, 0, tap_map).data_ptr()
                        self._conv(self.bwd, d, "deconv_dgrad_phase", f"{name}.p{py}{px}")
                        first = False
        self._deconv_wgrad(src, x, Cin, N, H, W, wkey, B)
        # bias gradient over the real channels only
        rows = B * 4 * H * W
        r = max(1, min(self.csum_rows, rows // 256))
        slot = self._rs_slot(r * doutC * 4)
        self.bwd.add("ksmi_channel_sum", lambda: (dout.data_ptr(), self.scr(slot), r, rows, doutC, self.dt),
                     self._elt_meta("channel_sum", rows * doutC))
        self._defer_rowsum(bkey, slot, 0, r, 1, 0, doutC, N)
        self._rs_tick()

