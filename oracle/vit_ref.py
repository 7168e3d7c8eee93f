"""Oracle (TEST INFRASTRUCTURE) for rows V1-V5 of SURVEY.md §8(a): FloodViT.

A functional, state-dict driven fp32 restatement on stock PyTorch-CPU ops of what the reference computes:

  ViT.to_patch_embedding   vision_transformer.py:121-126   -> _patch_embed
  Attention.forward        vision_transformer.py:50-66     -> _attention
  FeedForward              vision_transformer.py:19-32     -> _feed_forward
  Transformer.forward      vision_transformer.py:84-89     -> loop in floodvit_forward
  ViT.forward (pool False) vision_transformer.py:139-151   -> cls/pos, x[:, 1:]
  FinetunerSegmentation    model_utilities.py:80-94        -> rearrange b (h w) c -> b c h w
  Decoder.forward          model_utilities.py:36-48        -> _decoder
  `mlp` / default heads    model_utilities.py:59-72,88-93  -> _head_1x1 (bilinear to 224^2 FIRST, then the 1x1 convolutions)

Pinned to the real reference by tests/golden/floodvit_*.npz (oracle/gen_golden.py imports
/root/reference/models/vision_transformer.py and model_utilities.py here).  Only tests/, smoke() and
bench.py's cpu_baseline may import this module; the product path never does.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F


def floodvit_state_dict_spec(channels=6, image_size=224, patch_size=16, dim=1024, depth=24, heads=16, dim_head=64,
                             mlp_dim=2048, num_classes=3, head="decoder"):
    """Keys/shapes of FinetunerSegmentation(ViT(...), {'decoder': True}).state_dict() (mlp_head is nn.Identity); head = "mlp" /
    "linear": the configs {'mlp': True} / {'mlp': False, 'decoder': False} of model_utilities.py:59-72."""
    inner = heads * dim_head
    npatch = (image_size // patch_size) ** 2
    pd = channels * patch_size * patch_size
    s = OrderedDict()
    s["model.pos_embedding"] = (1, npatch + 1, dim)
    s["model.cls_token"] = (1, 1, dim)
    s["model.to_patch_embedding.1.weight"] = (pd,)
    s["model.to_patch_embedding.1.bias"] = (pd,)
    s["model.to_patch_embedding.2.weight"] = (dim, pd)
    s["model.to_patch_embedding.2.bias"] = (dim,)
    s["model.to_patch_embedding.3.weight"] = (dim,)
    s["model.to_patch_embedding.3.bias"] = (dim,)
    s["model.transformer.norm.weight"] = (dim,)
    s["model.transformer.norm.bias"] = (dim,)
    for i in range(depth):
        a, f = f"model.transformer.layers.{i}.0", f"model.transformer.layers.{i}.1"
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
    if head == "mlp":
        s["head.0.weight"] = (512, dim, 1, 1)
        s["head.0.bias"] = (512,)
        s["head.2.weight"] = (num_classes, 512, 1, 1)
        s["head.2.bias"] = (num_classes,)
        return s
    if head == "linear":
        s["head.weight"] = (num_classes, dim, 1, 1)
        s["head.bias"] = (num_classes,)
        return s
    s["head.deconv1.weight"] = (1024, 128, 4, 4)
    s["head.deconv1.bias"] = (128,)
    s["head.deconv2.weight"] = (128, 64, 4, 4)
    s["head.deconv2.bias"] = (64,)
    s["head.deconv3.weight"] = (64, num_classes, 4, 4)
    s["head.deconv3.bias"] = (num_classes,)
    return s


def new_state_dict(**hp):
    return OrderedDict((k, torch.zeros(shp)) for k, shp in floodvit_state_dict_spec(**hp).items())


def _ln(sd, key, x):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{key}.weight"], sd[f"{key}.bias"], 1e-5)


def _patch_embed(sd, img, p):
    B, Cc, H, W = img.shape
    h, w = H // p, W // p
    # "b c (h p1) (w p2) -> b (h w) (p1 p2 c)": the channel is the FASTEST index inside a patch
    x = img.reshape(B, Cc, h, p, w, p).permute(0, 2, 4, 3, 5, 1).reshape(B, h * w, p * p * Cc)
    k = "model.to_patch_embedding"
    x = _ln(sd, f"{k}.1", x)
    x = F.linear(x, sd[f"{k}.2.weight"], sd[f"{k}.2.bias"])
    return _ln(sd, f"{k}.3", x)


def _attention(sd, key, x, heads):
    B, N, _ = x.shape
    h = _ln(sd, f"{key}.norm", x)
    qkv = F.linear(h, sd[f"{key}.to_qkv.weight"])
    inner = qkv.shape[-1] // 3
    d = inner // heads
    q, k, v = (t.reshape(B, N, heads, d).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    dots = torch.matmul(q, k.transpose(-1, -2)) * d ** -0.5
    out = torch.matmul(dots.softmax(dim=-1), v).transpose(1, 2).reshape(B, N, inner)
    return F.linear(out, sd[f"{key}.to_out.0.weight"], sd[f"{key}.to_out.0.bias"])


def _feed_forward(sd, key, x):
    h = _ln(sd, f"{key}.net.0", x)
    h = F.gelu(F.linear(h, sd[f"{key}.net.1.weight"], sd[f"{key}.net.1.bias"]))
    return F.linear(h, sd[f"{key}.net.4.weight"], sd[f"{key}.net.4.bias"])


def _decoder(sd, x):
    x = F.relu(F.conv_transpose2d(x, sd["head.deconv1.weight"], sd["head.deconv1.bias"], stride=2, padding=1))
    x = F.interpolate(x, scale_factor=2)                      # nn.Upsample default mode = nearest
    x = F.relu(F.conv_transpose2d(x, sd["head.deconv2.weight"], sd["head.deconv2.bias"], stride=2, padding=1))
    return F.conv_transpose2d(x, sd["head.deconv3.weight"], sd["head.deconv3.bias"], stride=2, padding=1)


def _head_1x1(sd, x, size, masks=None, inter=None):
    """model_utilities.py:88-93: nn.Upsample(size, mode='bilinear') on the [B,1024,14,14] map, then `head` = Conv1x1 -> ReLU -> Conv1x1
    (`mlp`) or one Conv1x1 (default).  masks = {"h": bool} replaces the ReLU by x * mask (see floodvit_forward)."""
    x = F.interpolate(x, size=size, mode="bilinear")
    if "head.0.weight" in sd:
        h = F.conv2d(x, sd["head.0.weight"], sd["head.0.bias"])
        if inter is not None:
            inter["h"] = h
        h = F.relu(h) if masks is None else h * masks["h"]
        return F.conv2d(h, sd["head.2.weight"], sd["head.2.bias"])
    return F.conv2d(x, sd["head.weight"], sd["head.bias"])


def floodvit_forward(sd, img, heads, patch_size=16, return_tokens=False, inter=None, masks=None):
    """`inter` (optional dict) receives the intermediate activations the GPU tests compare against.  `masks` (optional
    {"d1": bool, "d2": bool}) replaces the two Decoder ReLUs by x * mask: a pre-activation within rounding distance of 0
    may land on either side on another device, and the backward comparison must use the same active set."""
    depth = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("model.transformer.layers."))
    x = _patch_embed(sd, img, patch_size)
    B, n, D = x.shape
    if inter is not None:
        inter["embed"] = x
    x = torch.cat((sd["model.cls_token"].expand(B, 1, D), x), dim=1) + sd["model.pos_embedding"][:, :n + 1]
    if inter is not None:
        inter["x0"] = x
    for i in range(depth):
        x = _attention(sd, f"model.transformer.layers.{i}.0", x, heads) + x
        x = _feed_forward(sd, f"model.transformer.layers.{i}.1", x) + x
        if inter is not None:
            inter[f"layer{i}"] = x
    x = _ln(sd, "model.transformer.norm", x)[:, 1:]
    if return_tokens:
        return x
    g = img.shape[2] // patch_size
    x = x.reshape(B, g, img.shape[3] // patch_size, D).permute(0, 3, 1, 2)
    if "head.deconv1.weight" not in sd:
        if inter is not None:
            inter["feat"] = x
        return _head_1x1(sd, x, tuple(img.shape[2:]), masks, inter)
    if inter is None and masks is None:
        return _decoder(sd, x)
    d1 = F.conv_transpose2d(x, sd["head.deconv1.weight"], sd["head.deconv1.bias"], stride=2, padding=1)
    u1 = F.interpolate(F.relu(d1) if masks is None else d1 * masks["d1"], scale_factor=2)
    d2 = F.conv_transpose2d(u1, sd["head.deconv2.weight"], sd["head.deconv2.bias"], stride=2, padding=1)
    d2 = F.relu(d2) if masks is None else d2 * masks["d2"]
    if inter is not None:
        inter.update(feat=x, d1=d1, u1=u1, d2=d2)
    return F.conv_transpose2d(d2, sd["head.deconv3.weight"], sd["head.deconv3.bias"], stride=2, padding=1)


def cross_entropy(logits, labels, weights=None):
    """create_loss 'cross_entropy' (utilities/utilities.py:307-321): nn.CrossEntropyLoss(weight, ignore_index=3)."""
    w = None if weights is None else torch.tensor(list(weights), dtype=logits.dtype)
    return F.cross_entropy(logits, labels, weight=w, ignore_index=3)


def loss_and_grads(sd, img, labels, heads, weights=None, masks=None):
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits = floodvit_forward(params, img, heads, masks=masks)
    loss = cross_entropy(logits, labels, weights)
    loss.backward()
    return logits.detach(), float(loss.detach()), {k: p.grad for k, p in params.items()}
