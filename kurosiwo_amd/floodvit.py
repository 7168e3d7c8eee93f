"""FloodViT (rows V1-V5 of SURVEY.md §8(a)) on hand-written gfx950 kernels.

Reference: /root/reference/models/vision_transformer.py:92-156 (lucidrains-style `ViT`: patchify ->
LN -> Linear -> LN, cls + pos, depth x [pre-norm Attention, FeedForward] with residuals, final LN,
returns x[:, 1:] when `pool` is False) wrapped by `FinetunerSegmentation` + the three-ConvTranspose2d
`Decoder` of /root/reference/models/model_utilities.py:22-94.

`ViT(...)` keeps the reference constructor and its 276 state-dict keys; it is the (picklable) encoder a
MAE pre-training run hands over.  `FinetunerSegmentation(encoder, configs)` is the hot path: it adopts
the encoder's weights under the reference's `model.*` names, adds `head.deconv{1,2,3}.*`, and runs
forward/backward as a static launch plan (kurosiwo_amd/floodvit_plan.py).  No CPU fallback.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from .arena import ArenaModule, PlanFn, _numel
from .runtime import require_gpu


def _pair(t):
    return t if isinstance(t, tuple) else (t, t)


def vit_param_spec(patch_dim, num_patches, dim, depth, heads, dim_head, mlp_dim, num_classes=None, prefix=""):
    """state-dict keys/shapes in the registration order of vision_transformer.py:121-137 (Transformer.norm is
    registered before the layers, :72-73)."""
    inner = heads * dim_head
    s = OrderedDict()
    s["pos_embedding"] = (1, num_patches + 1, dim)
    s["cls_token"] = (1, 1, dim)
    s["to_patch_embedding.1.weight"] = (patch_dim,)
    s["to_patch_embedding.1.bias"] = (patch_dim,)
    s["to_patch_embedding.2.weight"] = (dim, patch_dim)
    s["to_patch_embedding.2.bias"] = (dim,)
    s["to_patch_embedding.3.weight"] = (dim,)
    s["to_patch_embedding.3.bias"] = (dim,)
    s["transformer.norm.weight"] = (dim,)
    s["transformer.norm.bias"] = (dim,)
    for i in range(depth):
        a, f = f"transformer.layers.{i}.0", f"transformer.layers.{i}.1"
        s[f"{a}.norm.weight"] = (dim,)
        s[f"{a}.norm.bias"] = (dim,)
        s[f"{a}.to_qkv.weight"] = (3 * inner, dim)
        s[f"{a}.to_out.0.weight"] = (dim, inner)
        s[f"{a}.to_out.0.bias"] = (dim,)
        s[f"{f}.net.0.weight"] = (dim,)
        s[f"{f}.net.0.bias"] = (dim,)
        s[f"{f}.net.1.weight"] = (mlp_dim, dim)
        s[f"{f}.net.1.bias"] = (mlp_dim,)
        s[f"{f}.net.4.weight"] = (dim, mlp_dim)
        s[f"{f}.net.4.bias"] = (dim,)
    if num_classes is not None:
        s["mlp_head.weight"] = (num_classes, dim)
        s["mlp_head.bias"] = (num_classes,)
    return OrderedDict((prefix + k, v) for k, v in s.items())


def _init_vit_(get, spec, prefix=""):
    """PyTorch default initialisation of the reference modules: LayerNorm (1, 0), Linear
    kaiming_uniform(a=sqrt 5) = U(+-1/sqrt(fan_in)) for weight and bias, randn pos/cls (:128-129)."""
    with torch.no_grad():
        for key, shp in spec.items():
            p = get(key)
            k = key[len(prefix):]
            if k in ("pos_embedding", "cls_token"):
                p.normal_()
            elif len(shp) == 2:
                p.uniform_(-1 / math.sqrt(shp[1]), 1 / math.sqrt(shp[1]))
            elif k.endswith("bias") and (prefix + k[:-4] + "weight") in spec and len(spec[prefix + k[:-4] + "weight"]) == 2:
                fan_in = spec[prefix + k[:-4] + "weight"][1]
                p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
            elif k.endswith("weight"):
                p.fill_(1.0)
            else:
                p.zero_()


class ViT(nn.Module):
    """Parameter-compatible encoder (same ctor as vision_transformer.py:93-108, same 276 keys).  It is a weight
    container: the MI355X forward/backward lives in FinetunerSegmentation (the path the reference fine-tunes,
    model_utilities.py:158-165); MAE pre-training is row N3 of SURVEY.md §8(f)."""

    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool="cls", channels=3,
                 dim_head=64, dropout=0.0, emb_dropout=0.0):
        super().__init__()
        ih, iw = _pair(image_size)
        ph, pw = _pair(patch_size)
        if ih % ph or iw % pw:
            raise ValueError("Image dimensions must be divisible by the patch size.")
        if pool not in ("cls", "mean"):
            raise ValueError("pool type must be either cls (cls token) or mean (mean pooling)")
        if dropout != 0.0 or emb_dropout != 0.0:
            raise NotImplementedError("dropout > 0 (the reference configs use 0, configs/method/mae/mae.json)")
        self.hp = dict(image_size=(ih, iw), patch_size=(ph, pw), channels=channels, dim=dim, depth=depth, heads=heads,
                       dim_head=dim_head, mlp_dim=mlp_dim, num_classes=num_classes)
        self.pool = pool
        spec = vit_param_spec(channels * ph * pw, (ih // ph) * (iw // pw), dim, depth, heads, dim_head, mlp_dim, num_classes)
        self._spec = spec
        for key, shp in spec.items():
            parts = key.split(".")
            mod = self
            for part in parts[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, nn.Module())
                mod = mod._modules[part]
            mod.register_parameter(parts[-1], nn.Parameter(torch.empty(shp)))
        sd = dict(self.named_parameters())
        _init_vit_(lambda k: sd[k], spec)
        self.mlp_head.in_features = dim

    def forward(self, img):
        raise _lib.KsmiError("kurosiwo_amd.floodvit.ViT is a weight container; wrap it in FinetunerSegmentation")


def _infer_hp(encoder, configs):
    if hasattr(encoder, "hp"):
        return dict(encoder.hp)
    sd = encoder.state_dict()
    dim = sd["pos_embedding"].shape[2]
    npatch = sd["pos_embedding"].shape[1] - 1
    patch_dim = sd["to_patch_embedding.1.weight"].shape[0]
    depth = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.layers."))
    inner = sd["transformer.layers.0.0.to_qkv.weight"].shape[0] // 3
    try:
        heads = int(encoder.transformer.layers[0][0].heads)
    except Exception:
        heads = inner // 64
    img = int((configs or {}).get("image_size", 224))
    g = int(round(math.sqrt(npatch)))
    p = img // g
    return dict(image_size=(img, img), patch_size=(p, p), channels=patch_dim // (p * p), dim=dim, depth=depth, heads=heads,
                dim_head=inner // heads, mlp_dim=sd["transformer.layers.0.1.net.1.weight"].shape[0], num_classes=None)


class FinetunerSegmentation(ArenaModule):
    """model_utilities.py:51-94: logits[B, num_classes, 224, 224] = head(rearrange(ViT(x)[:, 1:], "b (h w) c -> b c h w")) with the
    `decoder` head (:22-48, the FloodViT configuration of SURVEY.md §8 V5), the `mlp` head or the default 1x1 head (:59-72; both
    after a bilinear resize to the image size, :88-93)."""

    def __init__(self, encoder, configs=None, pool=False, precision="bf16"):
        super().__init__()
        configs = dict(configs or {})
        if pool:
            raise NotImplementedError("pool=True (linear head on the mean token) is not part of the FloodViT path")
        # model_utilities.py:59-72: configs["mlp"] wins, then configs["decoder"], else one 1x1 convolution
        self.head_kind = "mlp" if configs.get("mlp", False) else ("decoder" if configs.get("decoder", True) else "linear")
        self.configs, self.pool, self.precision = configs, pool, precision
        hp = _infer_hp(encoder, configs)
        self.hp = hp
        if hp["dim_head"] != 64:
            raise NotImplementedError("attention kernel is specialised for dim_head = 64 (vision_transformer.py:103)")
        if hp["dim"] != 1024 and self.head_kind == "decoder":
            raise ValueError("Decoder.deconv1 hard-codes 1024 input channels (model_utilities.py:27)")
        self.num_classes = int(configs.get("num_classes", 3))
        ph, pw = hp["patch_size"]
        ih, iw = hp["image_size"]
        self.grid = (ih // ph, iw // pw)
        spec = vit_param_spec(hp["channels"] * ph * pw, self.grid[0] * self.grid[1], hp["dim"], hp["depth"], hp["heads"],
                              hp["dim_head"], hp["mlp_dim"], None, prefix="model.")
        if self.head_kind == "mlp":                      # nn.Sequential(Conv2d(dim, 512, 1), ReLU, Conv2d(512, classes, 1))
            spec["head.0.weight"] = (512, hp["dim"], 1, 1)
            spec["head.0.bias"] = (512,)
            spec["head.2.weight"] = (self.num_classes, 512, 1, 1)
            spec["head.2.bias"] = (self.num_classes,)
        elif self.head_kind == "linear":                 # nn.Conv2d(dim, classes, 1)
            spec["head.weight"] = (self.num_classes, hp["dim"], 1, 1)
            spec["head.bias"] = (self.num_classes,)
        else:
            spec["head.deconv1.weight"] = (1024, 128, 4, 4)
            spec["head.deconv1.bias"] = (128,)
            spec["head.deconv2.weight"] = (128, 64, 4, 4)
            spec["head.deconv2.bias"] = (64,)
            spec["head.deconv3.weight"] = (64, self.num_classes, 4, 4)
            spec["head.deconv3.bias"] = (self.num_classes,)
        self._setup_arena(spec)
        # adopt the encoder weights; head: ConvTranspose2d default init (fan_in = weight.size(1) * k * k)
        esd = encoder.state_dict()
        with torch.no_grad():
            for key, shp in spec.items():
                p = self._p(key).view(shp)
                if key.startswith("model."):
                    p.copy_(esd[key[6:]])
                else:
                    w = spec[key.rsplit(".", 1)[0] + ".weight"]
                    bound = 1 / math.sqrt(w[1] * w[2] * w[3])        # kaiming_uniform(a = sqrt 5): fan_in = weight.size(1) * k * k
                    p.uniform_(-bound, bound)
        if configs.get("linear_eval", False):            # model_utilities.py:160-161
            for key in spec:
                if key.startswith("model."):
                    self._param_obj(key).requires_grad_(False)

    def plan(self, B, training, with_backward):
        self._ensure_arena()
        # the plan skips the encoder's backward when every encoder parameter is frozen (linear_eval): part of the key
        enc_trains = any(self._param_obj(k).requires_grad for k in self._pspec if k.startswith("model."))
        key = (B, self.act_dtype(), bool(training), bool(with_backward), enc_trains)
        if key not in self._plans:
            from .floodvit_plan import FloodViTPlan
            self._plans[key] = FloodViTPlan(self, B, self.act_dtype(), with_backward)
        return self._plans[key]

    def forward(self, x):
        require_gpu(x)
        ih, iw = self.hp["image_size"]
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.hp["channels"], ih, iw):
            raise ValueError(f"expected [B,{self.hp['channels']},{ih},{iw}], got {tuple(x.shape)}")
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        plan = self.plan(x.shape[0], self.training, want_grad)
        x = x.contiguous().float()
        if not want_grad:
            return plan.run_forward(x).clone()
        if self._anchor is None or self._anchor.device != x.device:
            self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
        return PlanFn.apply(self._anchor, self, plan, x)


def num_params(spec):
    return sum(_numel(s) for s in spec.values())
