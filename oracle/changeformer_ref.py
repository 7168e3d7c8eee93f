"""Oracle (TEST INFRASTRUCTURE) for rows C1-C8 of SURVEY.md §8(a): ChangeFormerV6.

A functional, state-dict driven fp32 restatement on stock PyTorch-CPU ops of /root/reference/models/changeformer.py:

  OverlapPatchEmbed        :251-292   -> _patch_embed
  Attention (SR)           :148-208   -> _attention
  Mlp + DWConv             :85-133    -> _mlp
  Block                    :211-248   -> _block            (DropPath / Dropout: see below)
  EncoderTransformer_v3    :339-468   -> encoder_features
  conv_diff/make_prediction :31-46    -> _conv_diff / _make_pred (conv -> ReLU -> BatchNorm ordering)
  DecoderTransformer_v3    :485-641   -> decoder
  ChangeFormerV6.forward   :666-676   -> changeformer_forward

Stochastic layers: ChangeFormerV6 hard-codes drop_rate = attn_drop = drop_path_rate = 0.1 (:651-653), identities in eval mode.
Train-mode vectors exist in two flavours: changeformer.npz / changeformer_slc.npz with every nn.Dropout.p and DropPath.drop_prob
set to 0 on the reference module, and changeformer_drop.npz with the layers ON, their Bernoulli draws taken from the
counter-based stream of oracle/rng_ref.py (`stream=` below; the reference's own modules are run with the same draws in
oracle/gen_golden.py:gen_changeformer_drop).

Pinned to the real reference by tests/golden/changeformer_*.npz.  Only tests/, smoke() and bench.py's cpu_baseline may
import this module.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

EMBED_DIMS = (64, 128, 320, 512)
DEPTHS = (3, 3, 4, 3)
NUM_HEADS = (1, 2, 4, 8)
SR_RATIOS = (8, 4, 2, 1)
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _bn_spec(s, name, c):
    s[f"{name}.weight"] = (c,)
    s[f"{name}.bias"] = (c,)
    s[f"{name}.running_mean"] = (c,)
    s[f"{name}.running_var"] = (c,)
    s[f"{name}.num_batches_tracked"] = ()


def changeformer_state_dict_spec(input_nc=2, output_nc=3, embed_dim=256):
    """Ordered {key: shape} of ChangeFormerV6(input_nc, output_nc, decoder_softmax, embed_dim).state_dict()."""
    s = OrderedDict()
    cin = input_nc
    for i, c in enumerate(EMBED_DIMS):
        p = f"Tenc_x2.patch_embed{i + 1}"
        s[f"{p}.proj.weight"] = (c, cin, 7, 7)
        s[f"{p}.proj.bias"] = (c,)
        s[f"{p}.norm.weight"] = (c,)
        s[f"{p}.norm.bias"] = (c,)
        cin = c
    for st, c in enumerate(EMBED_DIMS):
        for i in range(DEPTHS[st]):
            b = f"Tenc_x2.block{st + 1}.{i}"
            s[f"{b}.norm1.weight"] = (c,)
            s[f"{b}.norm1.bias"] = (c,)
            s[f"{b}.attn.q.weight"] = (c, c)
            s[f"{b}.attn.q.bias"] = (c,)
            s[f"{b}.attn.kv.weight"] = (2 * c, c)
            s[f"{b}.attn.kv.bias"] = (2 * c,)
            s[f"{b}.attn.proj.weight"] = (c, c)
            s[f"{b}.attn.proj.bias"] = (c,)
            if SR_RATIOS[st] > 1:
                r = SR_RATIOS[st]
                s[f"{b}.attn.sr.weight"] = (c, c, r, r)
                s[f"{b}.attn.sr.bias"] = (c,)
                s[f"{b}.attn.norm.weight"] = (c,)
                s[f"{b}.attn.norm.bias"] = (c,)
            s[f"{b}.norm2.weight"] = (c,)
            s[f"{b}.norm2.bias"] = (c,)
            s[f"{b}.mlp.fc1.weight"] = (4 * c, c)
            s[f"{b}.mlp.fc1.bias"] = (4 * c,)
            s[f"{b}.mlp.dwconv.dwconv.weight"] = (4 * c, 1, 3, 3)
            s[f"{b}.mlp.dwconv.dwconv.bias"] = (4 * c,)
            s[f"{b}.mlp.fc2.weight"] = (c, 4 * c)
            s[f"{b}.mlp.fc2.bias"] = (c,)
        s[f"Tenc_x2.norm{st + 1}.weight"] = (c,)
        s[f"Tenc_x2.norm{st + 1}.bias"] = (c,)
    E = embed_dim
    for i in (4, 3, 2, 1):
        s[f"TDec_x2.linear_c{i}.proj.weight"] = (E, EMBED_DIMS[i - 1])
        s[f"TDec_x2.linear_c{i}.proj.bias"] = (E,)
    for i in (4, 3, 2, 1):
        d = f"TDec_x2.diff_c{i}"
        s[f"{d}.0.weight"] = (E, 2 * E, 3, 3)
        s[f"{d}.0.bias"] = (E,)
        _bn_spec(s, f"{d}.2", E)
        s[f"{d}.3.weight"] = (E, E, 3, 3)
        s[f"{d}.3.bias"] = (E,)
    for i in (4, 3, 2, 1):
        d = f"TDec_x2.make_pred_c{i}"
        s[f"{d}.0.weight"] = (output_nc, E, 3, 3)
        s[f"{d}.0.bias"] = (output_nc,)
        _bn_spec(s, f"{d}.2", output_nc)
        s[f"{d}.3.weight"] = (output_nc, output_nc, 3, 3)
        s[f"{d}.3.bias"] = (output_nc,)
    s["TDec_x2.linear_fuse.0.weight"] = (E, 4 * E, 1, 1)
    s["TDec_x2.linear_fuse.0.bias"] = (E,)
    _bn_spec(s, "TDec_x2.linear_fuse.1", E)
    for name in ("convd2x", "dense_2x.0.conv1", "dense_2x.0.conv2", "convd1x", "dense_1x.0.conv1", "dense_1x.0.conv2"):
        k = 4 if name.startswith("convd") else 3
        s[f"TDec_x2.{name}.conv2d.weight"] = (E, E, k, k)
        s[f"TDec_x2.{name}.conv2d.bias"] = (E,)
    s["TDec_x2.change_probability.conv2d.weight"] = (output_nc, E, 3, 3)
    s["TDec_x2.change_probability.conv2d.bias"] = (output_nc,)
    return s


def new_state_dict(input_nc=2, output_nc=3, embed_dim=256):
    sd = OrderedDict()
    for k, shp in changeformer_state_dict_spec(input_nc, output_nc, embed_dim).items():
        sd[k] = torch.zeros(shp, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
    return sd


def is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def _ln(sd, key, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{key}.weight"], sd[f"{key}.bias"], eps)


def _bn(sd, key, x, training, new_stats):
    """nn.BatchNorm2d; in training mode the updated running statistics go to `new_stats` (the state dict is not mutated)."""
    if not training:
        return F.batch_norm(x, sd[f"{key}.running_mean"], sd[f"{key}.running_var"], sd[f"{key}.weight"], sd[f"{key}.bias"],
                            False, BN_MOMENTUM, BN_EPS)
    rm, rv = sd[f"{key}.running_mean"].detach().clone(), sd[f"{key}.running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[f"{key}.weight"], sd[f"{key}.bias"], True, BN_MOMENTUM, BN_EPS)
    if new_stats is not None:
        new_stats[f"{key}.running_mean"], new_stats[f"{key}.running_var"] = rm, rv
        new_stats[f"{key}.num_batches_tracked"] = sd[f"{key}.num_batches_tracked"] + 1
    return y


def _patch_embed(sd, key, x, stride):
    x = F.conv2d(x, sd[f"{key}.proj.weight"], sd[f"{key}.proj.bias"], stride=stride, padding=3)
    B, Cc, H, W = x.shape
    x = x.flatten(2).transpose(1, 2)
    return _ln(sd, f"{key}.norm", x, 1e-5), H, W              # OverlapPatchEmbed.norm = nn.LayerNorm default eps (:268)


def _attention(sd, key, x, H, W, heads, sr, drop=None):
    B, N, Cc = x.shape
    d = Cc // heads
    q = F.linear(x, sd[f"{key}.q.weight"], sd[f"{key}.q.bias"]).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    if sr > 1:
        x_ = x.permute(0, 2, 1).reshape(B, Cc, H, W)
        x_ = F.conv2d(x_, sd[f"{key}.sr.weight"], sd[f"{key}.sr.bias"], stride=sr).reshape(B, Cc, -1).permute(0, 2, 1)
        x_ = _ln(sd, f"{key}.norm", x_, 1e-5)                  # Attention.norm = nn.LayerNorm default eps (:167)
    else:
        x_ = x
    kv = F.linear(x_, sd[f"{key}.kv.weight"], sd[f"{key}.kv.bias"]).reshape(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(dim=-1)
    if drop is not None:                                       # attn_drop (:203): element ((b*heads + h)*N + q)*Nk + key
        attn = attn * drop("attn", attn.shape)
    x = (attn @ v).transpose(1, 2).reshape(B, N, Cc)
    x = F.linear(x, sd[f"{key}.proj.weight"], sd[f"{key}.proj.bias"])
    return x if drop is None else x * drop("proj", x.shape)    # proj_drop (:207)


def _mlp(sd, key, x, H, W, drop=None):
    B, N, _ = x.shape
    x = F.linear(x, sd[f"{key}.fc1.weight"], sd[f"{key}.fc1.bias"])
    Ch = x.shape[-1]
    x = x.transpose(1, 2).reshape(B, Ch, H, W)
    x = F.conv2d(x, sd[f"{key}.dwconv.dwconv.weight"], sd[f"{key}.dwconv.dwconv.bias"], padding=1, groups=Ch)
    x = F.gelu(x.flatten(2).transpose(1, 2))
    if drop is not None:
        x = x * drop("mlp1", x.shape)                          # Mlp.drop after the activation (:130)
    x = F.linear(x, sd[f"{key}.fc2.weight"], sd[f"{key}.fc2.bias"])
    return x if drop is None else x * drop("mlp2", x.shape)    # ... and after fc2 (:132)


def _block(sd, key, x, H, W, heads, sr, drop=None):
    """Block.forward :245-248: x + drop_path(attn(norm1(x))), x + drop_path(mlp(norm2(x)))"""
    a = _attention(sd, f"{key}.attn", _ln(sd, f"{key}.norm1", x, 1e-6), H, W, heads, sr, drop)
    x = x + (a if drop is None else a * drop("path_attn", (x.shape[0], 1, 1)))
    m = _mlp(sd, f"{key}.mlp", _ln(sd, f"{key}.norm2", x, 1e-6), H, W, drop)
    return x + (m if drop is None else m * drop("path_mlp", (x.shape[0], 1, 1)))


def _block_drop(stream, gi, sample0):
    """mask factory of block gi for the images sample0 .. sample0+B-1 of the 2B-image batch the HIP path runs (date 1 first):
    element indices are row-major positions in that batched tensor (rng_ref.py)"""
    from . import rng_ref as G
    sites = {"attn": (G.SITE_ATTN, stream.p_attn), "proj": (G.SITE_PROJ, stream.p_drop), "mlp1": (G.SITE_MLP1, stream.p_drop),
             "mlp2": (G.SITE_MLP2, stream.p_drop)}

    def f(name, shape):
        if name.startswith("path"):
            m = stream.path(gi, G.SITE_PATH_ATTN if name == "path_attn" else G.SITE_PATH_MLP, sample0, shape[0])
            return torch.from_numpy(m).reshape(shape)
        site, p = sites[name]
        per = 1
        for d in shape[1:]:
            per *= d
        return torch.from_numpy(stream.elements(gi, site, p, sample0 * per, tuple(shape)))
    return f


def encoder_features(sd, x, inter=None, stream=None, sample0=0):
    B = x.shape[0]
    outs = []
    gi = 0
    for st in range(4):
        t, H, W = _patch_embed(sd, f"Tenc_x2.patch_embed{st + 1}", x, 4 if st == 0 else 2)
        if inter is not None:
            inter[f"pe{st + 1}"] = t
        for i in range(DEPTHS[st]):
            t = _block(sd, f"Tenc_x2.block{st + 1}.{i}", t, H, W, NUM_HEADS[st], SR_RATIOS[st],
                       None if stream is None else _block_drop(stream, gi, sample0))
            gi += 1
            if inter is not None:
                inter[f"s{st + 1}b{i}"] = t
        t = _ln(sd, f"Tenc_x2.norm{st + 1}", t, 1e-6)
        x = t.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs


def _relu(x, masks, name):
    """ReLU, or x * mask when the caller pins the active set (GPU parity tests: a pre-activation within rounding distance of
    0 may land on either side on another device and the backward comparison must use the same active set)."""
    if masks is None or name not in masks:
        return F.relu(x)
    return x * masks[name]


def _conv_diff(sd, key, x, training, new_stats, masks=None):
    name = key.split(".")[-1]
    x = _relu(F.conv2d(x, sd[f"{key}.0.weight"], sd[f"{key}.0.bias"], padding=1), masks, f"{name}.0")
    x = _bn(sd, f"{key}.2", x, training, new_stats)
    return _relu(F.conv2d(x, sd[f"{key}.3.weight"], sd[f"{key}.3.bias"], padding=1), masks, f"{name}.3")


def _make_pred(sd, key, x, training, new_stats):
    x = F.relu(F.conv2d(x, sd[f"{key}.0.weight"], sd[f"{key}.0.bias"], padding=1))
    x = _bn(sd, f"{key}.2", x, training, new_stats)
    return F.conv2d(x, sd[f"{key}.3.weight"], sd[f"{key}.3.bias"], padding=1)


def _res_block(sd, key, x, masks=None):
    out = _relu(F.conv2d(x, sd[f"{key}.conv1.conv2d.weight"], sd[f"{key}.conv1.conv2d.bias"], padding=1), masks, key.split(".")[1])
    out = F.conv2d(out, sd[f"{key}.conv2.conv2d.weight"], sd[f"{key}.conv2.conv2d.bias"], padding=1) * 0.1
    return out + x


def decoder(sd, f1, f2, training=False, new_stats=None, decoder_softmax=True, inter=None, masks=None):
    D = "TDec_x2"
    size1 = f1[0].shape[2:]
    outputs, ups, prev = [], [], None
    for i in (4, 3, 2, 1):
        a, b = f1[i - 1], f2[i - 1]
        n, _, h, w = a.shape

        def lin(t):
            y = F.linear(t.flatten(2).transpose(1, 2), sd[f"{D}.linear_c{i}.proj.weight"], sd[f"{D}.linear_c{i}.proj.bias"])
            return y.permute(0, 2, 1).reshape(n, -1, h, w)
        c = _conv_diff(sd, f"{D}.diff_c{i}", torch.cat((lin(a), lin(b)), dim=1), training, new_stats, masks)
        if prev is not None:
            c = c + F.interpolate(prev, scale_factor=2, mode="bilinear")
        outputs.append(_make_pred(sd, f"{D}.make_pred_c{i}", c, training, new_stats))
        ups.append(c if i == 1 else F.interpolate(c, size=size1, mode="bilinear", align_corners=False))
        prev = c
        if inter is not None:
            inter[f"c{i}"] = c
    x = F.conv2d(torch.cat(ups, dim=1), sd[f"{D}.linear_fuse.0.weight"], sd[f"{D}.linear_fuse.0.bias"])
    x = _bn(sd, f"{D}.linear_fuse.1", x, training, new_stats)
    if inter is not None:
        inter["fuse"] = x
    x = F.conv_transpose2d(x, sd[f"{D}.convd2x.conv2d.weight"], sd[f"{D}.convd2x.conv2d.bias"], stride=2, padding=1)
    x = _res_block(sd, f"{D}.dense_2x.0", x, masks)
    if inter is not None:
        inter["dense_2x"] = x
    x = F.conv_transpose2d(x, sd[f"{D}.convd1x.conv2d.weight"], sd[f"{D}.convd1x.conv2d.bias"], stride=2, padding=1)
    x = _res_block(sd, f"{D}.dense_1x.0", x, masks)
    if inter is not None:
        inter["dense_1x"] = x
    outputs.append(F.conv2d(x, sd[f"{D}.change_probability.conv2d.weight"], sd[f"{D}.change_probability.conv2d.bias"], padding=1))
    return [torch.sigmoid(o) for o in outputs] if decoder_softmax else outputs


def changeformer_forward(sd, x1, x2, training=False, new_stats=None, decoder_softmax=True, inter=None, masks=None, stream=None):
    """Returns the list of 5 outputs [(B,3,7,7), (B,3,14,14), (B,3,28,28), (B,3,56,56), (B,3,224,224)] (for 224x224 input)."""
    i1 = {} if inter is not None else None
    i2 = {} if inter is not None else None
    st = stream if training else None                           # nn.Dropout / DropPath are identities in eval mode
    f1, f2 = encoder_features(sd, x1, i1, st, 0), encoder_features(sd, x2, i2, st, x1.shape[0])
    if inter is not None:
        inter.update({f"A.{k}": v for k, v in i1.items()})
        inter.update({f"B.{k}": v for k, v in i2.items()})
        for i in range(4):
            inter[f"A.f{i + 1}"], inter[f"B.f{i + 1}"] = f1[i], f2[i]
    return decoder(sd, f1, f2, training, new_stats, decoder_softmax, inter, masks)


def loss_and_grads(sd, x1, x2, labels, weights=(1.0, 1.0, 1.0), with_dice=True, masks=None, stream=None):
    """One train-mode forward/backward with the reference's CD criterion on output[-1] (cd_trainer:138-166)."""
    from .snunet_ref import torch_ce_dice
    params = {k: (v.detach().clone().requires_grad_(True) if not is_buffer(k) else v) for k, v in sd.items()}
    new_stats = {}
    outs = changeformer_forward(params, x1, x2, training=True, new_stats=new_stats, masks=masks, stream=stream)
    loss = torch_ce_dice(outs[-1], labels, weights, with_dice)
    total = loss[0] if isinstance(loss, (tuple, list)) else loss
    total.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items() if not is_buffer(k)}
    return [o.detach() for o in outs], float(total.detach()), grads, new_stats
