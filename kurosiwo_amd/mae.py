"""MAE pre-training step for FloodViT on hand-written gfx950 kernels.

Reference: /root/reference/models/mae.py:9-124 (lucidrains-style masked autoencoder around the `ViT` encoder of
models/vision_transformer.py): patchify -> LN/Linear/LN -> + position rows -> keep a random 25 % of the tokens -> encoder
transformer -> Linear to the decoder width -> re-insert mask tokens with decoder position rows -> decoder transformer ->
Linear to pixel values of the masked patches -> F.mse_loss against the raw masked patches.

The decoder is `vit_pytorch.vit.Transformer` in the reference (third-party, absent from /root/reference and this image); it is
the class the reference's in-tree models/vision_transformer.py:69-89 copies (pre-norm Attention / FeedForward pairs and a final
LayerNorm), so the same layer builder and the same `layers.{i}.0|1.*`, `norm.*` parameter names serve both transformers.

`MAE(encoder=ViT(...), decoder_dim=..., ...)` keeps the reference constructor; parameters live in one flat fp32 arena under
the reference state-dict names (`mask_token`, `encoder.*`, `enc_to_dec.*`, `decoder.*`, `decoder_pos_emb.weight`, `to_pixels.*`;
the reference additionally exposes the encoder's patch-embedding parameters a second time as `patch_to_emb.{0,1,2}.*`: those
aliases are written by `state_dict()` and ignored on load).  forward(img) returns the reconstruction loss; the random
permutation is drawn exactly as the reference does (`torch.rand(B, N).argsort(-1)` on the device).  No CPU fallback.
"""
import math
from collections import OrderedDict

import torch

from . import _lib
from .arena import ArenaModule, PlanFn
from .floodvit import _infer_hp, vit_param_spec
from .runtime import require_gpu


def mae_param_spec(hp, decoder_dim, decoder_depth, decoder_heads, decoder_dim_head):
    ph, pw = hp["patch_size"]
    ih, iw = hp["image_size"]
    npatch = (ih // ph) * (iw // pw)
    pd = hp["channels"] * ph * pw
    s = OrderedDict()
    s["mask_token"] = (decoder_dim,)
    s.update(vit_param_spec(pd, npatch, hp["dim"], hp["depth"], hp["heads"], hp["dim_head"], hp["mlp_dim"], hp.get("num_classes"),
                            prefix="encoder."))
    if hp["dim"] != decoder_dim:
        s["enc_to_dec.weight"] = (decoder_dim, hp["dim"])
        s["enc_to_dec.bias"] = (decoder_dim,)
    inner = decoder_heads * decoder_dim_head
    s["decoder.norm.weight"] = (decoder_dim,)
    s["decoder.norm.bias"] = (decoder_dim,)
    for i in range(decoder_depth):
        a, f = f"decoder.layers.{i}.0", f"decoder.layers.{i}.1"
        s[f"{a}.norm.weight"] = (decoder_dim,)
        s[f"{a}.norm.bias"] = (decoder_dim,)
        s[f"{a}.to_qkv.weight"] = (3 * inner, decoder_dim)
        s[f"{a}.to_out.0.weight"] = (decoder_dim, inner)
        s[f"{a}.to_out.0.bias"] = (decoder_dim,)
        s[f"{f}.net.0.weight"] = (decoder_dim,)
        s[f"{f}.net.0.bias"] = (decoder_dim,)
        s[f"{f}.net.1.weight"] = (decoder_dim * 4, decoder_dim)
        s[f"{f}.net.1.bias"] = (decoder_dim * 4,)
        s[f"{f}.net.4.weight"] = (decoder_dim, decoder_dim * 4)
        s[f"{f}.net.4.bias"] = (decoder_dim,)
    s["decoder_pos_emb.weight"] = (npatch, decoder_dim)
    s["to_pixels.weight"] = (pd, decoder_dim)
    s["to_pixels.bias"] = (pd,)
    return s


_ALIASES = {"patch_to_emb.0.weight": "encoder.to_patch_embedding.1.weight", "patch_to_emb.0.bias": "encoder.to_patch_embedding.1.bias",
            "patch_to_emb.1.weight": "encoder.to_patch_embedding.2.weight", "patch_to_emb.1.bias": "encoder.to_patch_embedding.2.bias",
            "patch_to_emb.2.weight": "encoder.to_patch_embedding.3.weight", "patch_to_emb.2.bias": "encoder.to_patch_embedding.3.bias"}


class MAE(ArenaModule):
    """models/mae.py:9-124.  `encoder` is a kurosiwo_amd.floodvit.ViT (or any module with the reference ViT's state dict)."""

    def __init__(self, *, encoder, decoder_dim, masking_ratio=0.75, decoder_depth=1, decoder_heads=8, decoder_dim_head=64,
                 precision="bf16", configs=None):
        super().__init__()
        if not (0 < masking_ratio < 1):
            raise AssertionError("masking ratio must be kept between 0 and 1")
        if getattr(encoder, "pool", "cls") != "cls":
            raise NotImplementedError("pool='mean' (models/mae.py:66-67 adds the whole position table) is not part of the pre-training path")
        self.masking_ratio, self.decoder_dim, self.precision = masking_ratio, decoder_dim, precision
        hp = _infer_hp(encoder, configs)
        if hp["dim_head"] != 64 or decoder_dim_head != 64:
            raise NotImplementedError("attention kernel is specialised for dim_head = 64")
        self.hp = hp
        self.dhp = dict(dim=decoder_dim, depth=decoder_depth, heads=decoder_heads, dim_head=decoder_dim_head, mlp_dim=decoder_dim * 4)
        ph, pw = hp["patch_size"]
        ih, iw = hp["image_size"]
        self.grid = (ih // ph, iw // pw)
        self.num_patches = self.grid[0] * self.grid[1]
        self.num_masked = int(masking_ratio * self.num_patches)
        spec = mae_param_spec(hp, decoder_dim, decoder_depth, decoder_heads, decoder_dim_head)
        self._setup_arena(spec)
        esd = encoder.state_dict()
        with torch.no_grad():
            for key, shp in spec.items():
                p = self._p(key).view(shp)
                if key.startswith("encoder."):
                    p.copy_(esd[key[8:]])
                elif key in ("mask_token", "decoder_pos_emb.weight"):
                    p.normal_()                                  # torch.randn (mae.py:41), nn.Embedding default N(0, 1)
                elif len(shp) == 2:
                    p.uniform_(-1 / math.sqrt(shp[1]), 1 / math.sqrt(shp[1]))
                elif key.endswith(".bias") and key[:-4] + "weight" in spec and len(spec[key[:-4] + "weight"]) == 2:
                    fan_in = spec[key[:-4] + "weight"][1]
                    p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
                elif key.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
        # mlp_head / cls_token of the encoder take no part in MAE.forward (mae.py:54-124): no gradient reaches them
        self._register_state_dict_hook(MAE._add_aliases)
        self._register_load_state_dict_pre_hook(MAE._drop_aliases)
        self.last_indices = None

    @staticmethod
    def _add_aliases(module, state_dict, prefix, local_metadata):
        # the reference registers patch_to_emb right after the encoder (mae.py:31-32): keep that key order
        items = list(state_dict.items())
        last_enc = max((i for i, (k, _) in enumerate(items) if k.startswith(prefix + "encoder.")), default=len(items) - 1)
        extra = [(prefix + a, state_dict[prefix + k]) for a, k in _ALIASES.items() if prefix + k in state_dict]
        state_dict.clear()
        for i, (k, v) in enumerate(items):
            state_dict[k] = v
            if i == last_enc:
                for ak, av in extra:
                    state_dict[ak] = av
        return state_dict

    @staticmethod
    def _drop_aliases(state_dict, prefix, *args):
        for alias in _ALIASES:
            state_dict.pop(prefix + alias, None)

    def plan(self, B, with_backward):
        self._ensure_arena()
        key = (B, self.act_dtype(), bool(with_backward))
        if key not in self._plans:
            from .mae_plan import MAEPlan
            self._plans[key] = MAEPlan(self, B, self.act_dtype(), with_backward)
        return self._plans[key]

    def forward(self, img, rand_indices=None):
        require_gpu(img)
        ih, iw = self.hp["image_size"]
        if img.dim() != 4 or tuple(img.shape[1:]) != (self.hp["channels"], ih, iw):
            raise ValueError(f"expected [B,{self.hp['channels']},{ih},{iw}], got {tuple(img.shape)}")
        B = img.shape[0]
        if rand_indices is None:                                 # mae.py:73
            rand_indices = torch.rand(B, self.num_patches, device=img.device).argsort(dim=-1)
        self.last_indices = rand_indices
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        plan = self.plan(B, want_grad)
        img = img.contiguous().float()
        if not want_grad:
            return plan.run_forward(img, rand_indices).clone().reshape(())
        if self._anchor is None or self._anchor.device != img.device:
            self._anchor = torch.zeros(1, device=img.device, requires_grad=True)
        return PlanFn.apply(self._anchor, self, plan, img, rand_indices).reshape(())


def build_mae(configs, precision="bf16", channels=None):
    """training/train_mae.py:140-163: ViT(**mae.json) wrapped by MAE(decoder_dim/depth/heads, masked_ratio)."""
    from .floodvit import ViT
    enc = ViT(image_size=configs["image_size"], patch_size=configs["patch_size"], num_classes=configs["num_classes"], dim=configs["dim"],
              depth=configs["depth"], heads=configs["heads"], mlp_dim=configs["mlp_dim"],
              channels=channels if channels is not None else configs.get("num_channels", 3))
    return MAE(encoder=enc, masking_ratio=configs["masked_ratio"], decoder_dim=configs["decoder_dim"], decoder_depth=configs["decoder_depth"],
               decoder_heads=configs.get("decoder_heads", 8), precision=precision)
