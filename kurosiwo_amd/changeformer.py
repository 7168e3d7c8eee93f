"""ChangeFormerV6 (rows C1-C8 of SURVEY.md §8(a)) on hand-written gfx950 kernels.

Drop-in for the reference class (/root/reference/models/changeformer.py:643-676): same constructor, the same 373
state-dict keys, `model(x1, x2) -> [p_c4, p_c3, p_c2, p_c1, cp]` (five sigmoid maps when `decoder_softmax`), train/eval
BatchNorm semantics with running statistics.

MI355X-first design: the shared encoder runs ONCE over the 2B images of both dates (no BatchNorm in the encoder,
so batching the siamese branches is exact); every Linear, the 7x7 patch-embed and the k=s spatial-reduction convolutions are
implicit/explicit GEMMs on the MFMA kernels; the decoder's 3x3 / transposed convolutions (86 % of the FLOPs) use the
LDS-DMA implicit-GEMM kernel with ReLU, BatchNorm statistics, the 0.1-scaled residual and the concat fused into its
operand load / epilogue.

Stochastic layers: the reference hard-codes Dropout(0.1) / attention dropout 0.1 / DropPath(linspace(0, 0.1, 13)) (:651-653),
active in train mode.  They are applied here with the same probabilities and at the same places; the Bernoulli draws come from
a counter-based stream keyed by (seed, step, site, element) instead of torch's global generator (csrc/common.h, stochastic.hip),
so the backward pass regenerates the masks and oracle/rng_ref.py replays them: tests/golden/changeformer_drop.npz is the
reference's own modules run with their Dropout / DropPath draws replaced by that stream.  `drop_rate`, `attn_drop`,
`drop_path_rate` are attributes (set them to 0 for a deterministic regulariser-free run); `manual_seed(seed, step)` pins the stream.
Only the gradient of the last output (`cp`) is propagated: the reference trainer's default `multi_scale_train: false`
(configs/method/changeformer/changeformer.json, change_detection_trainer.py:138-166).
"""
import math
from collections import OrderedDict

import torch

from . import _lib
from .arena import ArenaModule, _numel
from .runtime import require_gpu

EMBED_DIMS = (64, 128, 320, 512)
DEPTHS = (3, 3, 4, 3)
NUM_HEADS = (1, 2, 4, 8)
SR_RATIOS = (8, 4, 2, 1)


def changeformer_specs(input_nc, output_nc, embed_dim):
    """(params, fp32 buffers, int64 counters) keyed and ordered as the reference's state dict."""
    p, b, c = OrderedDict(), OrderedDict(), OrderedDict()

    def bn(name, ch):
        p[f"{name}.weight"] = (ch,)
        p[f"{name}.bias"] = (ch,)
        b[f"{name}.running_mean"] = (ch,)
        b[f"{name}.running_var"] = (ch,)
        c[f"{name}.num_batches_tracked"] = ()
    cin = input_nc
    for i, ch in enumerate(EMBED_DIMS):
        k = f"Tenc_x2.patch_embed{i + 1}"
        p[f"{k}.proj.weight"] = (ch, cin, 7, 7)
        p[f"{k}.proj.bias"] = (ch,)
        p[f"{k}.norm.weight"] = (ch,)
        p[f"{k}.norm.bias"] = (ch,)
        cin = ch
    for st, ch in enumerate(EMBED_DIMS):
        for i in range(DEPTHS[st]):
            k = f"Tenc_x2.block{st + 1}.{i}"
            p[f"{k}.norm1.weight"] = (ch,)
            p[f"{k}.norm1.bias"] = (ch,)
            p[f"{k}.attn.q.weight"] = (ch, ch)
            p[f"{k}.attn.q.bias"] = (ch,)
            p[f"{k}.attn.kv.weight"] = (2 * ch, ch)
            p[f"{k}.attn.kv.bias"] = (2 * ch,)
            p[f"{k}.attn.proj.weight"] = (ch, ch)
            p[f"{k}.attn.proj.bias"] = (ch,)
            if SR_RATIOS[st] > 1:
                r = SR_RATIOS[st]
                p[f"{k}.attn.sr.weight"] = (ch, ch, r, r)
                p[f"{k}.attn.sr.bias"] = (ch,)
                p[f"{k}.attn.norm.weight"] = (ch,)
                p[f"{k}.attn.norm.bias"] = (ch,)
            p[f"{k}.norm2.weight"] = (ch,)
            p[f"{k}.norm2.bias"] = (ch,)
            p[f"{k}.mlp.fc1.weight"] = (4 * ch, ch)
            p[f"{k}.mlp.fc1.bias"] = (4 * ch,)
            p[f"{k}.mlp.dwconv.dwconv.weight"] = (4 * ch, 1, 3, 3)
            p[f"{k}.mlp.dwconv.dwconv.bias"] = (4 * ch,)
            p[f"{k}.mlp.fc2.weight"] = (ch, 4 * ch)
            p[f"{k}.mlp.fc2.bias"] = (ch,)
        p[f"Tenc_x2.norm{st + 1}.weight"] = (ch,)
        p[f"Tenc_x2.norm{st + 1}.bias"] = (ch,)
    E = embed_dim
    for i in (4, 3, 2, 1):
        p[f"TDec_x2.linear_c{i}.proj.weight"] = (E, EMBED_DIMS[i - 1])
        p[f"TDec_x2.linear_c{i}.proj.bias"] = (E,)
    for i in (4, 3, 2, 1):
        d = f"TDec_x2.diff_c{i}"
        p[f"{d}.0.weight"] = (E, 2 * E, 3, 3)
        p[f"{d}.0.bias"] = (E,)
        bn(f"{d}.2", E)
        p[f"{d}.3.weight"] = (E, E, 3, 3)
        p[f"{d}.3.bias"] = (E,)
    for i in (4, 3, 2, 1):
        d = f"TDec_x2.make_pred_c{i}"
        p[f"{d}.0.weight"] = (output_nc, E, 3, 3)
        p[f"{d}.0.bias"] = (output_nc,)
        bn(f"{d}.2", output_nc)
        p[f"{d}.3.weight"] = (output_nc, output_nc, 3, 3)
        p[f"{d}.3.bias"] = (output_nc,)
    p["TDec_x2.linear_fuse.0.weight"] = (E, 4 * E, 1, 1)
    p["TDec_x2.linear_fuse.0.bias"] = (E,)
    bn("TDec_x2.linear_fuse.1", E)
    for name in ("convd2x", "dense_2x.0.conv1", "dense_2x.0.conv2", "convd1x", "dense_1x.0.conv1", "dense_1x.0.conv2"):
        k = 4 if name.startswith("convd") else 3
        p[f"TDec_x2.{name}.conv2d.weight"] = (E, E, k, k)
        p[f"TDec_x2.{name}.conv2d.bias"] = (E,)
    p["TDec_x2.change_probability.conv2d.weight"] = (output_nc, E, 3, 3)
    p["TDec_x2.change_probability.conv2d.bias"] = (output_nc,)
    return p, b, c


class ChangeFormerV6(ArenaModule):
    def __init__(self, input_nc=3, output_nc=2, decoder_softmax=False, embed_dim=256, precision="bf16"):
        super().__init__()
        if output_nc > 8:
            raise _lib.KsmiError("ChangeFormerV6 (HIP): output_nc <= 8")
        if embed_dim % 32:
            raise _lib.KsmiError("ChangeFormerV6 (HIP): embed_dim must be a multiple of 32")
        self.input_nc, self.output_nc, self.decoder_softmax, self.embedding_dim = input_nc, output_nc, bool(decoder_softmax), embed_dim
        self.embed_dims, self.depths = list(EMBED_DIMS), list(DEPTHS)
        self.precision = precision
        self.drop_rate, self.attn_drop, self.drop_path_rate = 0.1, 0.1, 0.1       # ChangeFormerV6.__init__ :651-653
        ps, bs, cs = changeformer_specs(input_nc, output_nc, embed_dim)
        self._setup_arena(ps, bs, cs)
        self._init_parameters()

    def _init_parameters(self):
        """Encoder: _init_weights (:396-409) -- Linear trunc_normal(std .02), bias 0; LayerNorm (1, 0); Conv2d
        normal(0, sqrt(2 / fan_out)) with fan_out = k*k*Cout/groups, bias 0.  Decoder: PyTorch defaults (kaiming_uniform(a=sqrt 5)
        = U(+-1/sqrt(fan_in)) for weights and biases; BatchNorm (1, 0), running (0, 1))."""
        with torch.no_grad():
            for key, shp in self._pspec.items():
                p = self._p(key).view(shp)
                enc = key.startswith("Tenc_x2.")
                if len(shp) == 1:
                    is_norm = ".norm" in key or (not enc and self._is_bn(key))
                    if key.endswith("weight") and is_norm:
                        p.fill_(1.0)
                    elif enc or is_norm:
                        p.zero_()
                    else:
                        w = self._pspec[key[:-4] + "weight"]
                        fan_in = w[0] * w[2] * w[3] if (len(w) == 4 and "convd" in key) else _numel(w[1:])
                        p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
                elif enc and len(shp) == 2:
                    torch.nn.init.trunc_normal_(p, std=0.02)
                elif enc:
                    groups = shp[0] if shp[1] == 1 and "dwconv" in key else 1
                    p.normal_(0, math.sqrt(2.0 / (shp[2] * shp[3] * shp[0] // groups)))
                else:
                    fan_in = shp[0] * shp[2] * shp[3] if (len(shp) == 4 and "convd" in key) else _numel(shp[1:])
                    p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
            for key in self._bspec:
                self._b(key).fill_(1.0 if key.endswith("running_var") else 0.0)

    def _is_bn(self, key):
        return (key.rsplit(".", 1)[0] + ".running_mean") in self._bspec

    def plan(self, B, H, W, training, with_backward):
        self._ensure_arena()
        key = (B, H, W, self.act_dtype(), bool(training), bool(with_backward),
               (self.drop_rate, self.attn_drop, self.drop_path_rate) if training else None)
        if key not in self._plans:
            from .changeformer_plan import ChangeFormerPlan
            self._plans[key] = ChangeFormerPlan(self, B, H, W, self.act_dtype(), training, with_backward)
        return self._plans[key]

    def forward(self, x1, x2):
        require_gpu(x1)
        if x1.shape != x2.shape or x1.dim() != 4 or x1.shape[1] != self.input_nc:
            raise ValueError(f"expected two [B,{self.input_nc},H,W] tensors, got {tuple(x1.shape)} {tuple(x2.shape)}")
        B, _, H, W = x1.shape
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        plan = self.plan(B, H, W, self.training, want_grad)
        x1, x2 = x1.contiguous().float(), x2.contiguous().float()
        if not want_grad:
            plan.run_forward(x1, x2)
            return [o.clone() for o in plan.outputs]
        if self._anchor is None or self._anchor.device != x1.device:
            self._anchor = torch.zeros(1, device=x1.device, requires_grad=True)
        return list(_ChangeFormerFn.apply(self._anchor, x1, x2, self, plan))


class _ChangeFormerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x1, x2, model, plan):
        from .arena import stamp_forward
        ctx.model, ctx.plan = model, plan
        ctx.gen = stamp_forward(plan)
        ctx.set_materialize_grads(False)
        plan.run_forward(x1, x2)
        return tuple(o.clone() for o in plan.outputs)

    @staticmethod
    def backward(ctx, *douts):
        model, plan = ctx.model, ctx.plan
        if any(d is not None for d in douts[:-1]):
            raise NotImplementedError("ChangeFormerV6 (HIP): only the last output (cp) is differentiable (multi_scale_train = false)")
        if douts[-1] is None:
            return (None,) * 5
        from .arena import check_forward_stamp
        check_forward_stamp(plan, ctx.gen)
        model._check_no_grads()
        plan.run_backward(douts[-1].contiguous().float())
        model._attach_grads()
        return (None,) * 5
