"""BIT-CD as the reference ships it (row N2 of SURVEY.md §8): `configs/method/bit-cd/bit_cd.json` selects `net_G = "base_resnet18"`, i.e.
`define_G` builds `ResNet(input_nc, output_nc=3, output_sigmoid=False)` (/root/reference/models/bit_cd.py:686-688, 715-797): a siamese
ResNet-18 whose layer3 / layer4 keep stride 1 (`replace_stride_with_dilation=[False, True, True]`; the BasicBlock of that file resets
the dilation to 1, :97-98, so every 3x3 convolution is dense), nearest x2 upsampling, a 3x3 `conv_pred` to 32 channels per date,
|f1 - f2|, bilinear x4 upsampling and the two-layer classifier (conv3x3 -> BN -> ReLU -> conv3x3).  `BASE_Transformer` (:802-934,
the other three `net_G` values of define_G): the same backbone cut after layer3, a semantic tokenizer, a token encoder and a
per-date token decoder in front of the same head (kernels: csrc/bitcd.hip, plan: bitcd_plan.BitCDTransformerPlan).

Drop-in: same constructor, the same state-dict keys in the same order (including the unused `resnet.fc.*` of the torchvision-style
backbone: they receive no gradient; the fused optimizer still applies its weight decay to them, which the reference's optimizer skips
for `grad is None` -- no effect on any output), train / eval BatchNorm semantics with the shared backbone normalising each date on
its own.  Runs on the ResNet / decoder kernels of the U1 row (kurosiwo_amd/unet_plan.py).
"""
from collections import OrderedDict

import torch

from . import _lib
from .arena import ArenaModule, PlanFn
from .runtime import require_gpu

LAYERS = ((64, 1), (128, 2), (256, 1), (512, 1))      # (planes, stride of the first block): layer3/4 "dilated" = stride 1


def bitcd_specs(input_nc, output_nc):
    p, b, c = OrderedDict(), OrderedDict(), OrderedDict()

    def bn(name, ch):
        p[f"{name}.weight"] = (ch,)
        p[f"{name}.bias"] = (ch,)
        b[f"{name}.running_mean"] = (ch,)
        b[f"{name}.running_var"] = (ch,)
        c[f"{name}.num_batches_tracked"] = ()
    p["resnet.conv1.weight"] = (64, input_nc, 7, 7)
    bn("resnet.bn1", 64)
    cin = 64
    for li, (ch, stride) in enumerate(LAYERS):
        for bi in range(2):
            k = f"resnet.layer{li + 1}.{bi}"
            p[f"{k}.conv1.weight"] = (ch, cin, 3, 3)
            bn(f"{k}.bn1", ch)
            p[f"{k}.conv2.weight"] = (ch, ch, 3, 3)
            bn(f"{k}.bn2", ch)
            if bi == 0 and (stride != 1 or cin != ch):
                p[f"{k}.downsample.0.weight"] = (ch, cin, 1, 1)
                bn(f"{k}.downsample.1", ch)
            cin = ch
    p["resnet.fc.weight"] = (1000, 512)
    p["resnet.fc.bias"] = (1000,)
    p["classifier.0.weight"] = (32, 32, 3, 3)
    bn("classifier.1", 32)
    p["classifier.3.weight"] = (output_nc, 32, 3, 3)
    p["classifier.3.bias"] = (output_nc,)
    p["conv_pred.weight"] = (32, 512, 3, 3)
    p["conv_pred.bias"] = (32,)
    return p, b, c


class ResNet(ArenaModule):
    """bit_cd.py:715-797 (`base_resnet18`)"""

    def __init__(self, input_nc, output_nc, resnet_stages_num=5, backbone="resnet18", output_sigmoid=False, if_upsample_2x=True,
                 precision="bf16", init_gain=0.02):
        super().__init__()
        if backbone != "resnet18" or resnet_stages_num != 5 or output_sigmoid or not if_upsample_2x:
            raise NotImplementedError("BIT-CD (HIP): the shipped configuration only (base_resnet18: resnet18, 5 stages, x2 upsampling, logits)")
        if output_nc > 8:
            raise _lib.KsmiError("BIT-CD (HIP): output_nc <= 8")
        self.input_nc, self.output_nc, self.precision = input_nc, output_nc, precision
        ps, bs, cs = bitcd_specs(input_nc, output_nc)
        self._setup_arena(ps, bs, cs)
        with torch.no_grad():             # init_weights(net, 'normal', 0.02) (:654-683): conv / linear N(0, gain), BN weight N(1, gain), biases 0
            for key, shp in self._pspec.items():
                v = self._p(key).view(shp)
                if key.endswith("bias"):
                    v.zero_()
                elif len(shp) == 1:
                    v.normal_(1.0, init_gain)
                else:
                    v.normal_(0.0, init_gain)
            for key in self._bspec:
                self._b(key).fill_(1.0 if key.endswith("running_var") else 0.0)

    def plan(self, B, H, W, training, with_backward):
        self._ensure_arena()
        key = (B, H, W, self.act_dtype(), bool(training), bool(with_backward))
        if key not in self._plans:
            from .bitcd_plan import BitCDPlan
            self._plans[key] = BitCDPlan(self, B, H, W, self.act_dtype(), training, with_backward)
        return self._plans[key]

    def forward(self, x1, x2):
        require_gpu(x1)
        if x1.shape != x2.shape or x1.dim() != 4 or x1.shape[1] != self.input_nc or x1.shape[2] % 32 or x1.shape[3] % 32:
            raise ValueError(f"expected two [B,{self.input_nc},H,W] tensors with H, W multiples of 32, got {tuple(x1.shape)} {tuple(x2.shape)}")
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        plan = self.plan(x1.shape[0], x1.shape[2], x1.shape[3], self.training, want_grad)
        x1, x2 = x1.contiguous().float(), x2.contiguous().float()
        if not want_grad:
            return plan.run_forward(x1, x2).clone()
        if self._anchor is None or self._anchor.device != x1.device:
            self._anchor = torch.zeros(1, device=x1.device, requires_grad=True)
        return PlanFn.apply(self._anchor, self, plan, x1, x2)


def transformer_specs(input_nc, output_nc, token_len, enc_depth, dec_depth, dim_head, decoder_dim_head, with_pos):
    """parameters in the reference's registration order: the root's own pos_embedding, resnet, classifier, conv_pred, conv_a,
    transformer, transformer_decoder (bit_cd.py:806-855; Transformer :564-572, TransformerDecoder :581-591)"""
    p0, b, c = bitcd_specs(input_nc, output_nc)
    p0["conv_pred.weight"] = (32, 256, 3, 3)              # resnet_stages_num = 4 (:753-754)
    p = OrderedDict()
    if with_pos:
        p["pos_embedding"] = (1, 2 * token_len, 32)
    p.update(p0)
    p["conv_a.weight"] = (token_len, 32, 1, 1)

    def ff(f):
        p[f"{f}.norm.weight"], p[f"{f}.norm.bias"] = (32,), (32,)
        p[f"{f}.fn.net.0.weight"], p[f"{f}.fn.net.0.bias"] = (64, 32), (64,)
        p[f"{f}.fn.net.3.weight"], p[f"{f}.fn.net.3.bias"] = (32, 64), (32,)
    for li in range(enc_depth):
        a = f"transformer.layers.{li}.0.fn"
        p[f"{a}.norm.weight"], p[f"{a}.norm.bias"] = (32,), (32,)
        p[f"{a}.fn.to_qkv.weight"] = (3 * 8 * dim_head, 32)
        p[f"{a}.fn.to_out.0.weight"], p[f"{a}.fn.to_out.0.bias"] = (32, 8 * dim_head), (32,)
        ff(f"transformer.layers.{li}.1.fn")
    for li in range(dec_depth):
        a = f"transformer_decoder.layers.{li}.0.fn"
        p[f"{a}.norm.weight"], p[f"{a}.norm.bias"] = (32,), (32,)
        for w in ("to_q", "to_k", "to_v"):
            p[f"{a}.fn.{w}.weight"] = (8 * decoder_dim_head, 32)
        p[f"{a}.fn.to_out.0.weight"], p[f"{a}.fn.to_out.0.bias"] = (32, 8 * decoder_dim_head), (32,)
        ff(f"transformer_decoder.layers.{li}.1.fn")
    return p, b, c


class BASE_Transformer(ResNet):
    """bit_cd.py:802-934 with the settings define_G uses: tokenizer, token_trans, with_decoder, decoder_softmax, no decoder position
    table, x2 upsampling.  Anything else raises."""

    def __init__(self, input_nc, output_nc, with_pos, resnet_stages_num=5, token_len=4, token_trans=True, enc_depth=1, dec_depth=1, dim_head=64,
                 decoder_dim_head=64, tokenizer=True, if_upsample_2x=True, pool_mode="max", pool_size=2, backbone="resnet18", decoder_softmax=True,
                 with_decoder_pos=None, with_decoder=True, precision="bf16", init_gain=0.02):
        ArenaModule.__init__(self)
        if (backbone != "resnet18" or resnet_stages_num != 4 or not tokenizer or not token_trans or not with_decoder or not decoder_softmax
                or with_decoder_pos is not None or not if_upsample_2x or with_pos not in (None, "learned")):
            raise NotImplementedError("BIT-CD (HIP): BASE_Transformer as define_G builds it (resnet18, 4 stages, tokenizer, token encoder and decoder)")
        if token_len != 4 or dim_head < 1 or decoder_dim_head < 1 or enc_depth < 1 or dec_depth < 1:
            raise _lib.KsmiError("BIT-CD (HIP): token_len 4 (csrc/bitcd.hip), depths >= 1")
        if output_nc > 8:
            raise _lib.KsmiError("BIT-CD (HIP): output_nc <= 8")
        self.input_nc, self.output_nc, self.precision = input_nc, output_nc, precision
        self.with_pos, self.token_len, self.enc_depth, self.dec_depth = with_pos, token_len, enc_depth, dec_depth
        self.dim_head, self.decoder_dim_head = dim_head, decoder_dim_head
        ps, bs, cs = transformer_specs(input_nc, output_nc, token_len, enc_depth, dec_depth, dim_head, decoder_dim_head, with_pos)
        self._setup_arena(ps, bs, cs)
        with torch.no_grad():
            # nn.Parameter(torch.randn(...)) for the position table (:836), nn.LayerNorm defaults (weight 1, bias 0: init_weights only
            # touches Conv / Linear / BatchNorm2d, :654-683), then init_weights as for the base network
            for key, shp in self._pspec.items():
                v = self._p(key).view(shp)
                if key == "pos_embedding":
                    v.normal_(0.0, 1.0)
                elif ".norm." in key:
                    v.fill_(1.0 if key.endswith("weight") else 0.0)
                elif key.endswith("bias"):
                    v.zero_()
                elif len(shp) == 1:
                    v.normal_(1.0, init_gain)
                else:
                    v.normal_(0.0, init_gain)
            for key in self._bspec:
                self._b(key).fill_(1.0 if key.endswith("running_var") else 0.0)

    def plan(self, B, H, W, training, with_backward):
        self._ensure_arena()
        key = (B, H, W, self.act_dtype(), bool(training), bool(with_backward))
        if key not in self._plans:
            from .bitcd_plan import BitCDTransformerPlan
            self._plans[key] = BitCDTransformerPlan(self, B, H, W, self.act_dtype(), training, with_backward)
        return self._plans[key]


def define_G(args, in_channels, precision="bf16"):
    """bit_cd.py:686-707"""
    if args.get("init_type", "normal") != "normal":
        raise NotImplementedError("BIT-CD (HIP): init_type normal")
    gain = args.get("init_gain", 0.02)
    if args["net_G"] == "base_resnet18":
        return ResNet(input_nc=in_channels, output_nc=3, output_sigmoid=False, precision=precision, init_gain=gain)
    variants = {"base_transformer_pos_s4": {}, "base_transformer_pos_s4_dd8": dict(enc_depth=1, dec_depth=8),
                "base_transformer_pos_s4_dd8_dedim8": dict(enc_depth=1, dec_depth=8, decoder_dim_head=8)}
    if args["net_G"] in variants:
        return BASE_Transformer(input_nc=in_channels, output_nc=3, token_len=4, resnet_stages_num=4, with_pos="learned", precision=precision,
                                init_gain=gain, **variants[args["net_G"]])
    raise NotImplementedError("Generator model name [%s] is not recognized" % args["net_G"])
