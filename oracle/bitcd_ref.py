"""Oracle (TEST INFRASTRUCTURE) for row N2 of SURVEY.md §8: BIT-CD, the shipped configuration `net_G = base_resnet18` and the three
`BASE_Transformer` variants of define_G (bit_cd.py:686-707; second half of this file).

Functional, state-dict driven fp32 restatement on stock PyTorch-CPU ops of /root/reference/models/bit_cd.py: `ResNet.forward` :763-778,
`forward_single` :780-797 (conv1 7x7 s2 -> bn1 -> relu -> maxpool 3x3 s2 -> layer1..4 with strides (1, 2, 1, 1): layer3 / layer4 are
"dilated" but `BasicBlock` resets the dilation, :97-98 -> nearest x2 -> conv_pred 3x3), |f1 - f2|, bilinear x4 (nn.Upsample default
align_corners=False), `TwoLayerConv2d` classifier :416-424.  The shared backbone sees each date on its own (two BatchNorm updates per
step, date 1 first).  Pinned to the real reference by tests/golden/bitcd.npz.  Only tests/, smoke() and bench.py's cpu_baseline may import it.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

LAYERS = ((64, 1), (128, 2), (256, 1), (512, 1))
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def state_dict_spec(input_nc=2, output_nc=3):
    s = OrderedDict()

    def bn(name, ch):
        s[f"{name}.weight"], s[f"{name}.bias"] = (ch,), (ch,)
        s[f"{name}.running_mean"], s[f"{name}.running_var"], s[f"{name}.num_batches_tracked"] = (ch,), (ch,), ()
    s["resnet.conv1.weight"] = (64, input_nc, 7, 7)
    bn("resnet.bn1", 64)
    cin = 64
    for li, (ch, stride) in enumerate(LAYERS):
        for bi in range(2):
            k = f"resnet.layer{li + 1}.{bi}"
            s[f"{k}.conv1.weight"] = (ch, cin, 3, 3)
            bn(f"{k}.bn1", ch)
            s[f"{k}.conv2.weight"] = (ch, ch, 3, 3)
            bn(f"{k}.bn2", ch)
            if bi == 0 and (stride != 1 or cin != ch):
                s[f"{k}.downsample.0.weight"] = (ch, cin, 1, 1)
                bn(f"{k}.downsample.1", ch)
            cin = ch
    s["resnet.fc.weight"], s["resnet.fc.bias"] = (1000, 512), (1000,)
    s["classifier.0.weight"] = (32, 32, 3, 3)
    bn("classifier.1", 32)
    s["classifier.3.weight"], s["classifier.3.bias"] = (output_nc, 32, 3, 3), (output_nc,)
    s["conv_pred.weight"], s["conv_pred.bias"] = (32, 512, 3, 3), (32,)
    return s


def new_state_dict(input_nc=2, output_nc=3):
    sd = OrderedDict()
    for k, shp in state_dict_spec(input_nc, output_nc).items():
        sd[k] = torch.zeros(shp, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
    return sd


def is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def _bn(sd, key, x, training, stats):
    if not training:
        return F.batch_norm(x, sd[f"{key}.running_mean"], sd[f"{key}.running_var"], sd[f"{key}.weight"], sd[f"{key}.bias"], False, 0.0, BN_EPS)
    rm = stats.get(f"{key}.running_mean", sd[f"{key}.running_mean"]).detach().clone()
    rv = stats.get(f"{key}.running_var", sd[f"{key}.running_var"]).detach().clone()
    y = F.batch_norm(x, rm, rv, sd[f"{key}.weight"], sd[f"{key}.bias"], True, BN_MOMENTUM, BN_EPS)
    stats[f"{key}.running_mean"], stats[f"{key}.running_var"] = rm, rv
    stats[f"{key}.num_batches_tracked"] = stats.get(f"{key}.num_batches_tracked", sd[f"{key}.num_batches_tracked"]) + 1
    return y


def _block(sd, k, x, stride, training, stats):
    out = F.relu(_bn(sd, f"{k}.bn1", F.conv2d(x, sd[f"{k}.conv1.weight"], None, stride, 1), training, stats))
    out = _bn(sd, f"{k}.bn2", F.conv2d(out, sd[f"{k}.conv2.weight"], None, 1, 1), training, stats)
    if f"{k}.downsample.0.weight" in sd:
        x = _bn(sd, f"{k}.downsample.1", F.conv2d(x, sd[f"{k}.downsample.0.weight"], None, stride, 0), training, stats)
    return F.relu(out + x)


def forward_single(sd, x, training, stats, inter=None, tag=""):
    x = F.relu(_bn(sd, "resnet.bn1", F.conv2d(x, sd["resnet.conv1.weight"], None, 2, 3), training, stats))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (_, stride) in enumerate(LAYERS):
        for bi in range(2):
            x = _block(sd, f"resnet.layer{li + 1}.{bi}", x, stride if bi == 0 else 1, training, stats)
        if inter is not None:
            inter[f"layer{li + 1}{tag}"] = x
    x = F.interpolate(x, scale_factor=2)                                   # nn.Upsample(scale_factor=2): nearest
    x = F.conv2d(x, sd["conv_pred.weight"], sd["conv_pred.bias"], 1, 1)
    if inter is not None:
        inter[f"pred{tag}"] = x
    return x


def forward(sd, x1, x2, training=False, stats=None, inter=None):
    stats = {} if stats is None else stats
    f1 = forward_single(sd, x1, training, stats, inter, "_1")
    f2 = forward_single(sd, x2, training, stats, inter, "_2")
    x = torch.abs(f1 - f2)
    x = F.interpolate(x, scale_factor=4, mode="bilinear", align_corners=False)
    x = F.relu(_bn(sd, "classifier.1", F.conv2d(x, sd["classifier.0.weight"], None, 1, 1), training, stats))
    if inter is not None:
        inter["cls"] = x
    return F.conv2d(x, sd["classifier.3.weight"], sd["classifier.3.bias"], 1, 1)


def loss_and_grads(sd, x1, x2, labels, weights=(1.0, 1.0, 1.0)):
    from .snunet_ref import torch_ce_dice
    params = {k: (v.detach().clone().requires_grad_(True) if not is_buffer(k) else v) for k, v in sd.items()}
    stats = {}
    out = forward(params, x1, x2, True, stats)
    total = torch_ce_dice(out, labels, weights, True)
    total.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items() if not is_buffer(k)}
    return out.detach(), float(total.detach()), grads, stats


# =================================================================================================================================
# BASE_Transformer (bit_cd.py:802-934): ResNet-18 cut after layer3 (`resnet_stages_num=4`; layer4 and fc stay in the state dict, unused)
# -> nearest x2 -> conv_pred (256 -> 32) per date; semantic tokenizer (:857-865: conv_a 1x1 32 -> token_len, softmax over the PIXELS,
# tokens = attention-weighted sums of the feature map); the 2 x token_len tokens of both dates + a learned position table through
# `Transformer` (:564-578: pre-norm attention + feed-forward with residuals, scale = dim ** -0.5 with dim = 32, :531); per date the
# `TransformerDecoder` (:581-598): pixels attend to that date's tokens (`Cross_Attention` :476-524; ONE LayerNorm normalises both the
# pixels and the tokens, PreNorm2 :453-459) + feed-forward; |x1 - x2| -> bilinear x4 -> classifier.
# =================================================================================================================================
VARIANTS = {   # define_G (bit_cd.py:690-700)
    "base_transformer_pos_s4": dict(token_len=4, enc_depth=1, dec_depth=1, dim_head=64, decoder_dim_head=64),
    "base_transformer_pos_s4_dd8": dict(token_len=4, enc_depth=1, dec_depth=8, dim_head=64, decoder_dim_head=64),
    "base_transformer_pos_s4_dd8_dedim8": dict(token_len=4, enc_depth=1, dec_depth=8, dim_head=64, decoder_dim_head=8),
}
DIM, HEADS, MLP = 32, 8, 64
LN_EPS = 1e-5


def transformer_state_dict_spec(net_G, input_nc=2, output_nc=3):
    """keys in the order of the reference module: the root's own parameter (pos_embedding) first, then the children in
    registration order (resnet, classifier, conv_pred, conv_a, transformer, transformer_decoder)"""
    v = VARIANTS[net_G]
    base = state_dict_spec(input_nc, output_nc)
    base["conv_pred.weight"] = (32, 256, 3, 3)                      # resnet_stages_num = 4 (:753-754)
    s = OrderedDict()
    s["pos_embedding"] = (1, 2 * v["token_len"], DIM)
    s.update(base)
    s["conv_a.weight"] = (v["token_len"], 32, 1, 1)
    for li in range(v["enc_depth"]):
        a, f = f"transformer.layers.{li}.0.fn", f"transformer.layers.{li}.1.fn"
        inner = HEADS * v["dim_head"]
        s[f"{a}.norm.weight"], s[f"{a}.norm.bias"] = (DIM,), (DIM,)
        s[f"{a}.fn.to_qkv.weight"] = (3 * inner, DIM)
        s[f"{a}.fn.to_out.0.weight"], s[f"{a}.fn.to_out.0.bias"] = (DIM, inner), (DIM,)
        s[f"{f}.norm.weight"], s[f"{f}.norm.bias"] = (DIM,), (DIM,)
        s[f"{f}.fn.net.0.weight"], s[f"{f}.fn.net.0.bias"] = (MLP, DIM), (MLP,)
        s[f"{f}.fn.net.3.weight"], s[f"{f}.fn.net.3.bias"] = (DIM, MLP), (DIM,)
    for li in range(v["dec_depth"]):
        a, f = f"transformer_decoder.layers.{li}.0.fn", f"transformer_decoder.layers.{li}.1.fn"
        inner = HEADS * v["decoder_dim_head"]
        s[f"{a}.norm.weight"], s[f"{a}.norm.bias"] = (DIM,), (DIM,)
        for w in ("to_q", "to_k", "to_v"):
            s[f"{a}.fn.{w}.weight"] = (inner, DIM)
        s[f"{a}.fn.to_out.0.weight"], s[f"{a}.fn.to_out.0.bias"] = (DIM, inner), (DIM,)
        s[f"{f}.norm.weight"], s[f"{f}.norm.bias"] = (DIM,), (DIM,)
        s[f"{f}.fn.net.0.weight"], s[f"{f}.fn.net.0.bias"] = (MLP, DIM), (MLP,)
        s[f"{f}.fn.net.3.weight"], s[f"{f}.fn.net.3.bias"] = (DIM, MLP), (DIM,)
    return s


def new_transformer_state_dict(net_G, input_nc=2, output_nc=3):
    sd = OrderedDict()
    for k, shp in transformer_state_dict_spec(net_G, input_nc, output_nc).items():
        sd[k] = torch.zeros(shp, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
    return sd


def _backbone_s4(sd, x, training, stats, inter=None, tag=""):
    """forward_single with resnet_stages_num = 4 (:780-797)"""
    x = F.relu(_bn(sd, "resnet.bn1", F.conv2d(x, sd["resnet.conv1.weight"], None, 2, 3), training, stats))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (_, stride) in enumerate(LAYERS[:3]):
        for bi in range(2):
            x = _block(sd, f"resnet.layer{li + 1}.{bi}", x, stride if bi == 0 else 1, training, stats)
    x = F.interpolate(x, scale_factor=2)
    x = F.conv2d(x, sd["conv_pred.weight"], sd["conv_pred.bias"], 1, 1)
    if inter is not None:
        inter[f"pred{tag}"] = x
    return x


def _ln(sd, key, x):
    return F.layer_norm(x, (DIM,), sd[f"{key}.weight"], sd[f"{key}.bias"], LN_EPS)


def _heads(t, h):
    b, n, _ = t.shape
    return t.view(b, n, h, -1).permute(0, 2, 1, 3)                      # 'b n (h d) -> b h n d'


def _ff(sd, f, x):
    y = _ln(sd, f"{f}.norm", x)
    y = F.linear(F.gelu(F.linear(y, sd[f"{f}.fn.net.0.weight"], sd[f"{f}.fn.net.0.bias"])), sd[f"{f}.fn.net.3.weight"], sd[f"{f}.fn.net.3.bias"])
    return y + x


def semantic_tokens(sd, x):
    """:857-865"""
    b, c, h, w = x.shape
    sa = torch.softmax(F.conv2d(x, sd["conv_a.weight"]).view(b, -1, h * w), dim=-1)
    return torch.einsum("bln,bcn->blc", sa, x.view(b, c, -1))


def token_encoder(sd, tokens, depth):
    """:879-883 + Transformer :564-578 / Attention :527-561"""
    x = tokens + sd["pos_embedding"]
    scale = DIM ** -0.5
    for li in range(depth):
        a, f = f"transformer.layers.{li}.0.fn", f"transformer.layers.{li}.1.fn"
        q, k, v = (_heads(t, HEADS) for t in F.linear(_ln(sd, f"{a}.norm", x), sd[f"{a}.fn.to_qkv.weight"]).chunk(3, dim=-1))
        attn = (torch.einsum("bhid,bhjd->bhij", q, k) * scale).softmax(dim=-1)
        out = torch.einsum("bhij,bhjd->bhid", attn, v).permute(0, 2, 1, 3).reshape(x.shape[0], x.shape[1], -1)
        x = F.linear(out, sd[f"{a}.fn.to_out.0.weight"], sd[f"{a}.fn.to_out.0.bias"]) + x
        x = _ff(sd, f, x)
    return x


def token_decoder(sd, x, m, depth):
    """:885-894 + TransformerDecoder :581-598 / Cross_Attention :476-524; x [b,c,h,w] pixels, m [b,l,c] tokens"""
    b, c, h, w = x.shape
    x = x.flatten(2).transpose(1, 2)
    scale = DIM ** -0.5
    for li in range(depth):
        a, f = f"transformer_decoder.layers.{li}.0.fn", f"transformer_decoder.layers.{li}.1.fn"
        xn, mn = _ln(sd, f"{a}.norm", x), _ln(sd, f"{a}.norm", m)
        q = _heads(F.linear(xn, sd[f"{a}.fn.to_q.weight"]), HEADS)
        k = _heads(F.linear(mn, sd[f"{a}.fn.to_k.weight"]), HEADS)
        v = _heads(F.linear(mn, sd[f"{a}.fn.to_v.weight"]), HEADS)
        attn = (torch.einsum("bhid,bhjd->bhij", q, k) * scale).softmax(dim=-1)
        out = torch.einsum("bhij,bhjd->bhid", attn, v).permute(0, 2, 1, 3).reshape(b, h * w, -1)
        x = F.linear(out, sd[f"{a}.fn.to_out.0.weight"], sd[f"{a}.fn.to_out.0.bias"]) + x
        x = _ff(sd, f, x)
    return x.transpose(1, 2).reshape(b, c, h, w)


def transformer_forward(sd, net_G, x1, x2, training=False, stats=None, inter=None):
    """BASE_Transformer.forward (:906-934) with tokenizer, token_trans, with_decoder (the define_G settings)"""
    v = VARIANTS[net_G]
    stats = {} if stats is None else stats
    f1 = _backbone_s4(sd, x1, training, stats, inter, "_1")
    f2 = _backbone_s4(sd, x2, training, stats, inter, "_2")
    t1, t2 = semantic_tokens(sd, f1), semantic_tokens(sd, f2)
    tokens = token_encoder(sd, torch.cat([t1, t2], dim=1), v["enc_depth"])
    t1, t2 = tokens.chunk(2, dim=1)
    if inter is not None:
        inter["tokens"] = tokens
    y1 = token_decoder(sd, f1, t1, v["dec_depth"])
    y2 = token_decoder(sd, f2, t2, v["dec_depth"])
    if inter is not None:
        inter["dec_1"], inter["dec_2"] = y1, y2
    x = torch.abs(y1 - y2)
    x = F.interpolate(x, scale_factor=4, mode="bilinear", align_corners=False)
    x = F.relu(_bn(sd, "classifier.1", F.conv2d(x, sd["classifier.0.weight"], None, 1, 1), training, stats))
    if inter is not None:
        inter["cls"] = x
    return F.conv2d(x, sd["classifier.3.weight"], sd["classifier.3.bias"], 1, 1)


def transformer_loss_and_grads(sd, net_G, x1, x2, labels, weights=(1.0, 1.0, 1.0)):
    from .snunet_ref import torch_ce_dice
    params = {k: (v.detach().clone().requires_grad_(True) if not is_buffer(k) else v) for k, v in sd.items()}
    stats = {}
    out = transformer_forward(params, net_G, x1, x2, True, stats)
    total = torch_ce_dice(out, labels, weights, True)
    total.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items() if not is_buffer(k)}
    return out.detach(), float(total.detach()), grads, stats
