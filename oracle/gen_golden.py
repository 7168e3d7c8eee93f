#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REAL REFERENCE (/root/reference).

Runs only in the build container (the reference never travels to the GPU box;
only these small data fixtures do).  Usage:  python oracle/gen_golden.py

What is produced (all inputs/weights are regenerated from oracle/seeded.py by
the tests, so only expected outputs are stored):

  loss_cases.npz      reference BCEandDiceLoss / nn.CrossEntropyLoss values + grads
  snunet_small.npz    SNUNet_ECAM(c, 3, base_channel=8) B=2 32x32: eval logits, train
                      logits, loss, per-parameter grad stats + selected full grads, BN
                      running stats after one step, 3-step Adam loss sequence
  snunet_full.npz     SNUNet_ECAM(2, 3, 32) 224x224: eval logits (B=1) subsample +
                      full argmax mask; train step (B=2) loss + grad norms
  floodvit_small.npz  FinetunerSegmentation(ViT(dim 1024, depth 2, heads 4, mlp 512, 6 ch), decoder head) B=2:
                      encoder tokens + logits subsamples, weighted-CE loss, per-parameter grad stats, selected grads
  changeformer.npz    ChangeFormerV6(2, 3, decoder_softmax=True, embed_dim=256): state-dict inventory, eval outputs (B=1), train-mode
                      (dropout / drop-path probabilities set to 0) outputs, ce+dice loss, grad stats, BN running statistics (B=2)
  floodvit_full.npz   the mae.json encoder (depth 24, heads 16, mlp 2048) B=1: logits subsample, argmax, loss, grad stats

models/model_utilities.py imports every model family of the reference plus three packages that are not installed
here (segmentation_models_pytorch, denoising_diffusion_pytorch, torchsummary, timm); they are irrelevant to the FloodViT
classes and are replaced by empty placeholder modules for the duration of that import (SURVEY.md §2, row "Model factory").
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle.seeded import seeded_fill_, seeded_labels, seeded_tensor  # noqa: E402

from models.snunet import SNUNet_ECAM  # noqa: E402  (reference)
from utilities.bce_and_dice import BCEandDiceLoss  # noqa: E402  (reference)

OUT = os.path.join(ROOT, "tests", "golden")
CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]  # utilities.py:393-397
FULL_GRAD_KEYS = [
    "conv0_0.conv1.weight", "conv0_0.conv1.bias", "conv0_0.bn1.weight", "conv0_0.bn2.bias",
    "conv1_0.conv2.weight", "conv4_0.bn2.weight", "Up1_0.up.weight", "Up4_0.up.bias",
    "conv0_4.conv1.weight", "conv2_2.conv1.bias", "ca.fc1.weight", "ca.fc2.weight",
    "ca1.fc1.weight", "ca1.fc2.weight", "conv_final.weight", "conv_final.bias",
]


def sar_like(name, shape):
    """Normalised-backscatter-like inputs: randn clipped to the VV/VH range of
    SURVEY.md §8(d) (after clamp 0.15 + normalise)."""
    return seeded_tensor(name, shape).mul_(1.0).clamp_(-2.23, 5.75)


def gen_loss():
    out = {}
    # KAT-loss-1
    kat_logits = torch.tensor([[[[1, -.5], [.25, 2]], [[0, .5], [-1, .5]], [[-1, 1.5], [.75, -2]]]], dtype=torch.float32)
    kat_lbl = torch.tensor([[[0, 2], [3, 1]]], dtype=torch.int64)
    cases = {"kat": (kat_logits, kat_lbl)}
    cases["rand"] = (seeded_tensor("loss.rand.logits", (2, 3, 16, 16)) * 2.0, seeded_labels("loss.rand.labels", (2, 16, 16)))
    cases["big"] = (seeded_tensor("loss.big.logits", (3, 3, 64, 48)) * 4.0, seeded_labels("loss.big.labels", (3, 64, 48), p_invalid=0.3))
    for cname, (x, t) in cases.items():
        for wname, w in (("unit", [1.0, 1.0, 1.0]), ("cw", CLASS_WEIGHTS)):
            crit = BCEandDiceLoss(weights=w, ignore_index=3, use_softmax=True)
            xx = x.clone().requires_grad_(True)
            dice = crit.dice(xx, t)
            ce = crit.bce(xx, t)
            total = crit(xx, t)
            total.backward()
            out[f"{cname}.{wname}.dice"] = dice.detach().numpy()
            out[f"{cname}.{wname}.ce"] = ce.detach().numpy()
            out[f"{cname}.{wname}.total"] = total.detach().numpy()
            out[f"{cname}.{wname}.grad"] = xx.grad.numpy().copy()
            # plain (weighted) cross entropy = create_loss 'cross_entropy' train mode
            xx2 = x.clone().requires_grad_(True)
            ce2 = torch.nn.CrossEntropyLoss(weight=torch.tensor(w), ignore_index=3)(xx2, t)
            ce2.backward()
            out[f"{cname}.{wname}.ce_only"] = ce2.detach().numpy()
            out[f"{cname}.{wname}.ce_only_grad"] = xx2.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "loss_cases.npz"), **out)
    print("loss_cases.npz", {k: float(v) for k, v in out.items() if v.ndim == 0 and k.startswith("kat")})


def _ref_model(c, bc):
    m = SNUNet_ECAM(c, 3, base_channel=bc)
    seeded_fill_(m.state_dict())
    return m


def _grad_stats(model):
    stats, full = {}, {}
    for k, p in model.named_parameters():
        g = p.grad.detach().double()
        stats[k] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
        if k in FULL_GRAD_KEYS:
            full[k] = p.grad.detach().numpy().copy()
    return stats, full


def gen_snunet_small():
    out = {}
    for c in (2, 3):
        tag = f"c{c}"
        B, H, W, bc = 2, 32, 32, 8
        xA = sar_like(f"small.{tag}.xA", (B, c, H, W))
        xB = sar_like(f"small.{tag}.xB", (B, c, H, W))
        lbl = seeded_labels(f"small.{tag}.lbl", (B, H, W))
        model = _ref_model(c, bc)
        model.eval()
        with torch.no_grad():
            out[f"{tag}.eval_logits"] = model(xA, xB).numpy().copy()
        # --- one train step, ce+dice with class weights, Adam lr 1e-3
        model = _ref_model(c, bc)
        model.train()
        crit = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        losses = []
        for step in range(3):
            opt.zero_grad()
            logits = model(xA, xB)
            loss = crit(logits, lbl)
            loss.backward()
            if step == 0:
                out[f"{tag}.train_logits"] = logits.detach().numpy().copy()
                stats, full = _grad_stats(model)
                for k, v in stats.items():
                    out[f"{tag}.gstat.{k}"] = v
                for k, v in full.items():
                    out[f"{tag}.grad.{k}"] = v
            opt.step()
            losses.append(float(loss))
            if step == 0:
                sd = model.state_dict()
                for k in ("conv0_0.bn1", "conv0_0.bn2", "conv4_0.bn1", "conv0_4.bn2", "conv2_1.bn1"):
                    out[f"{tag}.bn.{k}.running_mean"] = sd[f"{k}.running_mean"].numpy().copy()
                    out[f"{tag}.bn.{k}.running_var"] = sd[f"{k}.running_var"].numpy().copy()
                    out[f"{tag}.bn.{k}.num_batches_tracked"] = sd[f"{k}.num_batches_tracked"].numpy().copy()
                for k in ("conv0_0.conv1.weight", "conv_final.weight", "ca.fc1.weight"):
                    out[f"{tag}.param1.{k}"] = sd[k].numpy().copy()
        out[f"{tag}.losses"] = np.array(losses)
        sd = model.state_dict()
        out[f"{tag}.param3_sums"] = np.array([float(sd[k].double().sum()) for k in sd if sd[k].dtype.is_floating_point])
        print("snunet_small", tag, losses)
    np.savez_compressed(os.path.join(OUT, "snunet_small.npz"), **out)


def gen_snunet_full():
    out = {}
    c, bc, H, W = 2, 32, 224, 224
    xA = sar_like("full.xA", (1, c, H, W))
    xB = sar_like("full.xB", (1, c, H, W))
    model = _ref_model(c, bc)
    model.eval()
    with torch.no_grad():
        logits = model(xA, xB)
    out["eval_logits_sub"] = logits[:, :, ::8, ::8].numpy().copy()
    out["eval_argmax"] = logits.argmax(1).numpy().astype(np.uint8)
    top2 = logits.topk(2, dim=1).values
    out["eval_margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    out["eval_logits_absmax"] = np.array(float(logits.abs().max()))
    # train step, B=2
    xA = sar_like("full.train.xA", (2, c, H, W))
    xB = sar_like("full.train.xB", (2, c, H, W))
    lbl = seeded_labels("full.train.lbl", (2, H, W))
    model = _ref_model(c, bc)
    model.train()
    crit = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)
    logits = model(xA, xB)
    loss = crit(logits, lbl)
    loss.backward()
    out["train_logits_sub"] = logits[:, :, ::8, ::8].detach().numpy().copy()
    out["train_loss"] = np.array(float(loss))
    stats, full = _grad_stats(model)
    for k, v in stats.items():
        out[f"gstat.{k}"] = v
    for k in ("conv0_0.conv1.weight", "conv_final.weight", "ca.fc1.weight", "ca1.fc2.weight", "Up1_3.up.bias"):
        out[f"grad.{k}"] = full[k] if k in full else dict(model.named_parameters())[k].grad.numpy().copy()
    print("snunet_full loss", float(loss))
    np.savez_compressed(os.path.join(OUT, "snunet_full.npz"), **out)


def gen_snunet_bench():
    """The benchmarked configuration (BASELINE.json configs[1]: SNUNet-ECAM c=2 bc=32, batch 32, 224x224) on the REAL reference in
    fp32: train-mode logits (BatchNorm statistics over 32 tiles), ce+dice loss, gradient statistics.  Inputs are the synthetic SAR
    tiles of the benchmark (kurosiwo_amd/synthetic.make_batch) so the bf16 HIP path is checked on the data distribution it is timed on."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    out = {}
    c, bc, B = 2, 32, 32
    (xA, xB), lbl = cd_inputs(make_batch(B, 224, 224, seed=1234), ("pre_event_1", "post_event"))
    model = _ref_model(c, bc)
    model.train()
    crit = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)
    logits = model(xA, xB)
    loss = crit(logits, lbl)
    loss.backward()
    out["train_logits_sub"] = logits[:, :, ::16, ::16].detach().numpy().copy()
    out["train_logits_absmax"] = np.array(float(logits.detach().abs().max()))
    out["train_argmax_sub"] = logits[::8].detach().argmax(1).numpy().astype(np.uint8)
    top2 = logits[::8].detach().topk(2, dim=1).values
    out["train_margin_sub"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    out["train_loss"] = np.array(float(loss))
    stats, full = _grad_stats(model)
    for k, v in stats.items():
        out[f"gstat.{k}"] = v
    for k in ("conv_final.weight", "conv0_4.conv1.weight", "Up1_3.up.weight", "conv0_0.bn1.weight"):
        out[f"grad.{k}"] = dict(model.named_parameters())[k].grad.numpy().copy()
    sd = model.state_dict()
    for k in ("conv0_0.bn1", "conv0_4.bn2", "conv2_1.bn1"):
        out[f"bn.{k}.running_mean"] = sd[f"{k}.running_mean"].numpy().copy()
        out[f"bn.{k}.running_var"] = sd[f"{k}.running_var"].numpy().copy()
    print("snunet_bench loss", float(loss))
    np.savez_compressed(os.path.join(OUT, "snunet_bench.npz"), **out)


def _import_floodvit_reference():
    import types
    import models.upernet  # noqa: F401  (pulls in transformers before the placeholders exist)
    for name, attrs in (("segmentation_models_pytorch", ()), ("denoising_diffusion_pytorch", ("GaussianDiffusion", "Unet")),
                        ("torchsummary", ("summary",)), ("timm", ()), ("timm.models", ()),
                        ("timm.models.layers", ("DropPath", "to_2tuple", "trunc_normal_"))):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                import importlib.machinery
                mod = types.ModuleType(name)
                mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
                mod.__path__ = []
                for a in attrs:
                    setattr(mod, a, object)
                sys.modules[name] = mod
    from models.model_utilities import FinetunerSegmentation  # noqa: E402  (reference)
    from models.vision_transformer import ViT  # noqa: E402  (reference)
    return ViT, FinetunerSegmentation


FLOODVIT_SMALL = dict(channels=6, image_size=224, patch_size=16, dim=1024, depth=2, heads=4, mlp_dim=512)
FLOODVIT_FULL = dict(channels=6, image_size=224, patch_size=16, dim=1024, depth=24, heads=16, mlp_dim=2048)   # configs/method/mae/mae.json
FLOODVIT_GRAD_KEYS = [
    "model.cls_token", "model.to_patch_embedding.1.weight", "model.to_patch_embedding.2.bias", "model.to_patch_embedding.3.bias",
    "model.transformer.norm.weight", "model.transformer.layers.0.0.norm.weight", "model.transformer.layers.0.0.to_out.0.bias",
    "model.transformer.layers.1.1.net.1.bias", "model.transformer.layers.1.1.net.4.bias", "model.transformer.layers.0.1.net.0.bias",
    "head.deconv1.bias", "head.deconv2.bias", "head.deconv3.weight", "head.deconv3.bias",
]


def _ref_floodvit(hp, head="decoder"):
    ViT, FinetunerSegmentation = _import_floodvit_reference()
    enc = ViT(image_size=hp["image_size"], patch_size=hp["patch_size"], num_classes=1000, dim=hp["dim"], depth=hp["depth"],
              heads=hp["heads"], mlp_dim=hp["mlp_dim"], channels=hp["channels"])
    cfg = {"mlp": head == "mlp", "decoder": head == "decoder", "num_classes": 3, "image_size": 224, "finetuning_patch_size": hp["patch_size"]}
    model = FinetunerSegmentation(encoder=enc, configs=cfg)
    seeded_fill_(model.state_dict())
    return model


def gen_floodvit(tag, hp, B, head="decoder"):
    """head = "mlp" / "linear": the other two heads of FinetunerSegmentation (model_utilities.py:59-72) on the small encoder"""
    out = {}
    x = sar_like(f"floodvit.{tag}.x", (B, hp["channels"], 224, 224))
    lbl = seeded_labels(f"floodvit.{tag}.lbl", (B, 224, 224))
    model = _ref_floodvit(hp, head)
    model.train()
    out["state_dict_keys"] = np.array(list(model.state_dict().keys()))
    tokens = model.model(x)                                   # [B,196,1024] = ViT.forward with pool False
    out["tokens_sub"] = tokens[:, ::7, ::16].detach().numpy().copy()
    logits = model(x)
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor(CLASS_WEIGHTS), ignore_index=3)    # create_loss 'cross_entropy' train mode
    loss = crit(logits, lbl)
    loss.backward()
    out["logits_sub"] = logits[:, :, ::8, ::8].detach().numpy().copy()
    out["argmax"] = logits.argmax(1).numpy().astype(np.uint8)
    top2 = logits.topk(2, dim=1).values
    out["margin"] = (top2[:, 0] - top2[:, 1]).detach().numpy().astype(np.float16)
    out["loss"] = np.array(float(loss))
    for k, p in model.named_parameters():
        g = p.grad.detach().double()
        out[f"gstat.{k}"] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
        if k in FLOODVIT_GRAD_KEYS or (head != "decoder" and k.startswith("head.")):
            out[f"grad.{k}"] = p.grad.detach().numpy().copy()
    print(f"floodvit_{tag} loss", float(loss), "logits absmax", float(logits.abs().max()))
    np.savez_compressed(os.path.join(OUT, f"floodvit_{tag}.npz"), **out)


def gen_floodvit_bench():
    """BASELINE.json configs[4] per-GPU shard exactly as benchmarked (full-depth ViT d1024 L24 h16 mlp2048 + Decoder head, batch 16,
    224 x 224, the synthetic 3-date SAR tiles of bench.py) on the REAL reference in fp32: encoder tokens, logits, weighted-CE loss,
    gradient statistics of every parameter -- what the bf16 HIP path is held to at the size and on the data it is timed on."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from kurosiwo_amd.synthetic import make_batch, seg_inputs
    out = {}
    B = 16
    x, lbl = seg_inputs(make_batch(B, 224, 224, seed=1234))           # [post, pre1, pre2] = 3 dates x 2 ch
    model = _ref_floodvit(FLOODVIT_FULL, "decoder")
    model.train()
    tokens = model.model(x)
    out["tokens_sub"] = tokens[::4, ::7, ::16].detach().numpy().copy()
    out["tokens_absmax"] = np.array(float(tokens.detach().abs().max()))
    logits = model(x)
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor(CLASS_WEIGHTS), ignore_index=3)
    loss = crit(logits, lbl)
    loss.backward()
    out["logits_sub"] = logits[:, :, ::8, ::8].detach().numpy().copy()
    out["logits_absmax"] = np.array(float(logits.detach().abs().max()))
    out["argmax_sub"] = logits[::4].detach().argmax(1).numpy().astype(np.uint8)
    top2 = logits[::4].detach().topk(2, dim=1).values
    out["margin_sub"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    out["loss"] = np.array(float(loss))
    for k, p in model.named_parameters():
        g = p.grad.detach().double()
        out[f"gstat.{k}"] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
        if k in FLOODVIT_GRAD_KEYS or k in ("model.transformer.layers.23.1.net.4.bias", "model.transformer.layers.12.0.to_out.0.bias"):
            out[f"grad.{k}"] = p.grad.detach().numpy().copy()
    print("floodvit_bench loss", float(loss), "logits absmax", float(logits.abs().max()))
    np.savez_compressed(os.path.join(OUT, "floodvit_bench.npz"), **out)


MAE_SMALL = dict(channels=2, image_size=224, patch_size=16, dim=1024, depth=2, heads=4, mlp_dim=512, decoder_dim=512, decoder_depth=2,
                 decoder_heads=4)
MAE_GRAD_KEYS = ["mask_token", "encoder.to_patch_embedding.2.bias", "encoder.transformer.norm.weight", "encoder.transformer.layers.0.0.to_out.0.bias",
                 "encoder.transformer.layers.1.1.net.4.bias", "enc_to_dec.bias", "decoder.norm.weight", "decoder.layers.0.0.norm.weight",
                 "decoder.layers.1.1.net.1.bias", "to_pixels.bias"]


def gen_mae():
    """models/mae.py with `vit_pytorch.vit.Transformer` bound to the reference's in-tree copy of that class
    (models/vision_transformer.py:69-89): the third-party package is not vendored and not installed here."""
    import types
    ViT, _ = _import_floodvit_reference()
    import models.vision_transformer as vt
    if "vit_pytorch" not in sys.modules:
        import importlib.machinery
        pkg = types.ModuleType("vit_pytorch"); pkg.__spec__ = importlib.machinery.ModuleSpec("vit_pytorch", None); pkg.__path__ = []
        sub = types.ModuleType("vit_pytorch.vit"); sub.__spec__ = importlib.machinery.ModuleSpec("vit_pytorch.vit", None)
        sub.Transformer = vt.Transformer
        sys.modules["vit_pytorch"], sys.modules["vit_pytorch.vit"] = pkg, sub
    from models.mae import MAE  # noqa: E402  (reference)
    hp = MAE_SMALL
    B = 2
    enc = ViT(image_size=hp["image_size"], patch_size=hp["patch_size"], num_classes=1000, dim=hp["dim"], depth=hp["depth"], heads=hp["heads"],
              mlp_dim=hp["mlp_dim"], channels=hp["channels"])
    model = MAE(encoder=enc, masking_ratio=0.75, decoder_dim=hp["decoder_dim"], decoder_depth=hp["decoder_depth"], decoder_heads=hp["decoder_heads"])
    seeded_fill_(model.state_dict())
    model.train()
    x = sar_like("mae.small.x", (B, hp["channels"], 224, 224))
    out = {"state_dict_keys": np.array(list(model.state_dict().keys()))}
    torch.manual_seed(4242)
    idx = torch.rand(B, 196).argsort(dim=-1)                  # the draw MAE.forward makes first (mae.py:73)
    torch.manual_seed(4242)
    loss = model(x)
    loss.backward()
    out["rand_indices"] = idx.numpy()
    out["loss"] = np.array(float(loss))
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        gd = g.detach().double()
        out[f"gstat.{k}"] = np.array([float(gd.norm()), float(gd.sum()), float(gd.abs().max())])
        if k in MAE_GRAD_KEYS:
            out[f"grad.{k}"] = g.detach().numpy().copy()
    print("mae_small loss", float(loss))
    np.savez_compressed(os.path.join(OUT, "mae_small.npz"), **out)


def _import_changeformer_reference():
    """models/changeformer.py needs three helpers of timm (not installed): DropPath, to_2tuple, trunc_normal_.  The placeholders
    below stand in for them during the import: weights are seeded-filled afterwards (initialisation is irrelevant) and every
    stochastic layer is disabled for the vectors (eval mode, or p = 0), where timm's DropPath is the identity as well."""
    import importlib.machinery
    import types

    class DropPath(torch.nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if not (self.training and self.drop_prob > 0):
                return x
            # timm drop_path: x / keep_prob * Bernoulli(keep_prob) per sample; the draw comes from the counter-based stream
            # (gen_changeformer_drop) -- there is no golden vector with torch's own generator
            assert getattr(self, "scale_fn", None) is not None, "drop_path > 0 needs the counter-based stream"
            return x * self.scale_fn(x)
    attrs = {"timm": {}, "timm.models": {}, "timm.models.layers": {
        "DropPath": DropPath, "to_2tuple": lambda v: v if isinstance(v, tuple) else (v, v), "trunc_normal_": torch.nn.init.trunc_normal_}}
    for name, a in attrs.items():
        mod = types.ModuleType(name)
        mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
        mod.__path__ = []
        for k, v in a.items():
            setattr(mod, k, v)
        sys.modules[name] = mod
    from models.changeformer import ChangeFormerV6  # noqa: E402  (reference)
    return ChangeFormerV6


CHANGEFORMER_GRAD_KEYS = [
    "Tenc_x2.patch_embed1.proj.weight", "Tenc_x2.patch_embed1.norm.bias", "Tenc_x2.patch_embed3.proj.bias",
    "Tenc_x2.block1.0.attn.q.bias", "Tenc_x2.block1.0.attn.sr.bias", "Tenc_x2.block1.2.attn.norm.weight",
    "Tenc_x2.block2.1.attn.kv.bias", "Tenc_x2.block3.3.mlp.dwconv.dwconv.weight", "Tenc_x2.block3.0.mlp.dwconv.dwconv.bias",
    "Tenc_x2.block4.2.mlp.fc2.bias", "Tenc_x2.block4.0.norm1.weight", "Tenc_x2.norm2.weight", "Tenc_x2.norm4.bias",
    "TDec_x2.linear_c4.proj.bias", "TDec_x2.linear_c1.proj.weight", "TDec_x2.diff_c4.0.bias", "TDec_x2.diff_c1.2.weight",
    "TDec_x2.diff_c2.2.bias", "TDec_x2.diff_c3.3.bias", "TDec_x2.linear_fuse.0.bias", "TDec_x2.linear_fuse.1.weight",
    "TDec_x2.convd2x.conv2d.bias", "TDec_x2.dense_2x.0.conv2.conv2d.bias", "TDec_x2.dense_1x.0.conv1.conv2d.bias",
    "TDec_x2.change_probability.conv2d.weight", "TDec_x2.change_probability.conv2d.bias",
]


def gen_changeformer():
    ChangeFormerV6 = _import_changeformer_reference()
    out = {}
    c = 2

    def ref_model():
        m = ChangeFormerV6(input_nc=c, output_nc=3, decoder_softmax=True, embed_dim=256)   # model_utilities.py:198-204
        seeded_fill_(m.state_dict())
        return m
    model = ref_model()
    sd = model.state_dict()
    out["state_dict_keys"] = np.array(list(sd.keys()))
    out["state_dict_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    # ---- eval forward, B = 1
    x1 = sar_like("changeformer.eval.x1", (1, c, 224, 224))
    x2 = sar_like("changeformer.eval.x2", (1, c, 224, 224))
    model.eval()
    with torch.no_grad():
        outs = model(x1, x2)
        feats = model.Tenc_x2(x1)
    for i, o in enumerate(outs[:4]):
        out[f"eval.out{i}"] = o.numpy().copy()
    out["eval.out4_sub"] = outs[4][:, :, ::8, ::8].numpy().copy()
    out["eval.argmax"] = outs[4].argmax(1).numpy().astype(np.uint8)
    top2 = outs[4].topk(2, dim=1).values
    out["eval.margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    for i, f in enumerate(feats):
        out[f"eval.feat{i + 1}_sub"] = f[:, ::4, ::2, ::2].numpy().copy()
    # ---- train forward/backward, B = 2, stochastic layers off (p = 0)
    x1 = sar_like("changeformer.train.x1", (2, c, 224, 224))
    x2 = sar_like("changeformer.train.x2", (2, c, 224, 224))
    lbl = seeded_labels("changeformer.train.lbl", (2, 224, 224))
    model = ref_model()
    model.train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    crit = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)
    outs = model(x1, x2)
    loss = crit(outs[-1], lbl)                      # change_detection_trainer.py:138-166 (multi_scale_train false)
    loss.backward()
    for i, o in enumerate(outs[:4]):
        out[f"train.out{i}"] = o.detach().numpy().copy()
    out["train.out4_sub"] = outs[4][:, :, ::8, ::8].detach().numpy().copy()
    out["train.loss"] = np.array(float(loss.detach()))
    for k, p in model.named_parameters():
        if p.grad is None:
            out[f"gstat.{k}"] = np.zeros(3)
            continue
        g = p.grad.detach().double()
        out[f"gstat.{k}"] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
        if k in CHANGEFORMER_GRAD_KEYS:
            out[f"grad.{k}"] = p.grad.detach().numpy().copy()
    sd = model.state_dict()
    for k in ("TDec_x2.diff_c4.2", "TDec_x2.diff_c1.2", "TDec_x2.make_pred_c2.2", "TDec_x2.linear_fuse.1"):
        out[f"bn.{k}.running_mean"] = sd[f"{k}.running_mean"].numpy().copy()
        out[f"bn.{k}.running_var"] = sd[f"{k}.running_var"].numpy().copy()
        out[f"bn.{k}.num_batches_tracked"] = sd[f"{k}.num_batches_tracked"].numpy().copy()
    print("changeformer train loss", float(loss.detach()), "keys", len(sd))
    np.savez_compressed(os.path.join(OUT, "changeformer.npz"), **out)


def gen_changeformer_slc():
    """BASELINE.json configs[3] as written: SLC tiles, 4 bands per date (dataset/Dataset.py:986-1228; utilities/utilities.py:386-390
    doubles num_channels) -> ChangeFormerV6(input_nc=4).  Small fixture: eval outputs, train loss, gradient statistics, the first
    patch-embedding gradient (the only tensor whose shape depends on input_nc)."""
    ChangeFormerV6 = _import_changeformer_reference()
    out = {}
    c = 4

    def ref_model():
        m = ChangeFormerV6(input_nc=c, output_nc=3, decoder_softmax=True, embed_dim=256)
        seeded_fill_(m.state_dict())
        return m
    model = ref_model()
    sd = model.state_dict()
    out["state_dict_keys"] = np.array(list(sd.keys()))
    out["state_dict_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    x1 = sar_like("changeformer.slc.eval.x1", (1, c, 224, 224))
    x2 = sar_like("changeformer.slc.eval.x2", (1, c, 224, 224))
    model.eval()
    with torch.no_grad():
        outs = model(x1, x2)
    for i, o in enumerate(outs[:4]):
        out[f"eval.out{i}"] = o.numpy().copy()
    out["eval.out4_sub"] = outs[4][:, :, ::8, ::8].numpy().copy()
    out["eval.argmax"] = outs[4].argmax(1).numpy().astype(np.uint8)
    top2 = outs[4].topk(2, dim=1).values
    out["eval.margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    x1 = sar_like("changeformer.slc.train.x1", (2, c, 224, 224))
    x2 = sar_like("changeformer.slc.train.x2", (2, c, 224, 224))
    lbl = seeded_labels("changeformer.slc.train.lbl", (2, 224, 224))
    model = ref_model()
    model.train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    crit = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)
    outs = model(x1, x2)
    loss = crit(outs[-1], lbl)
    loss.backward()
    out["train.out4_sub"] = outs[4][:, :, ::8, ::8].detach().numpy().copy()
    out["train.loss"] = np.array(float(loss.detach()))
    for k, p in model.named_parameters():
        if p.grad is None:                     # (parameters outside the graph of output[-1]: the multi-scale prediction heads)
            out[f"gstat.{k}"] = np.zeros(3)
            continue
        g = p.grad.detach().double()
        out[f"gstat.{k}"] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
    out["grad.Tenc_x2.patch_embed1.proj.weight"] = dict(model.named_parameters())["Tenc_x2.patch_embed1.proj.weight"].grad.detach().numpy().copy()
    print("changeformer slc train loss", float(loss.detach()), "keys", len(sd))
    np.savez_compressed(os.path.join(OUT, "changeformer_slc.npz"), **out)


DROP_SEED, DROP_STEP = 20240607, 1


def _changeformer_drop_model(c, B):
    """the REFERENCE ChangeFormerV6 in train mode with every nn.Dropout / DropPath drawing from the counter-based stream (see
    gen_changeformer_drop); B = images per date"""
    from oracle import rng_ref as G
    ChangeFormerV6 = _import_changeformer_reference()
    model = ChangeFormerV6(input_nc=c, output_nc=3, decoder_softmax=True, embed_dim=256)
    seeded_fill_(model.state_dict())
    model.train()
    stream = G.DropStream(DROP_SEED, DROP_STEP, model.drop_rate, model.attn_drop, model.drop_path_rate)
    depths = [3, 3, 4, 3]
    ndrop = 0

    def install(mod, gi, roles, per_pass, path):
        calls = {"n": 0}

        def scale(x):
            k = calls["n"]
            calls["n"] += 1
            sample0 = (k // per_pass) * B                       # pass 0 = date 1, pass 1 = date 2
            role = roles[k % per_pass]
            if path:
                m = stream.path(gi, role, sample0, x.shape[0])
                return torch.from_numpy(m).reshape(-1, *([1] * (x.dim() - 1)))
            per = int(np.prod(x.shape[1:]))
            p = stream.p_attn if role == G.SITE_ATTN else stream.p_drop
            return torch.from_numpy(stream.elements(gi, role, p, sample0 * per, tuple(x.shape)))
        if path:
            mod.scale_fn = scale
        else:
            assert isinstance(mod, torch.nn.Dropout) and abs(mod.p - 0.1) < 1e-12
            mod.forward = lambda x: x * scale(x)
    gi = 0
    for st in range(4):
        for i in range(depths[st]):
            blk = getattr(model.Tenc_x2, f"block{st + 1}")[i]
            install(blk.attn.attn_drop, gi, [G.SITE_ATTN], 1, False)
            install(blk.attn.proj_drop, gi, [G.SITE_PROJ], 1, False)
            install(blk.mlp.drop, gi, [G.SITE_MLP1, G.SITE_MLP2], 2, False)
            ndrop += 3
            if gi > 0:                                          # dpr[0] = 0 -> nn.Identity (:222)
                assert abs(blk.drop_path.drop_prob - stream.dpr[gi]) < 1e-7, (blk.drop_path.drop_prob, stream.dpr[gi])
                install(blk.drop_path, gi, [G.SITE_PATH_ATTN, G.SITE_PATH_MLP], 2, True)
            gi += 1
    assert ndrop == sum(isinstance(m, torch.nn.Dropout) for m in model.modules()), "an nn.Dropout outside the encoder blocks"
    return model


def gen_changeformer_drop():
    """Train-mode step of the REFERENCE ChangeFormerV6 with its stochastic layers ON (drop_rate = attn_drop = drop_path_rate = 0.1,
    changeformer.py:651-653).  The module graph, the places and probabilities of every nn.Dropout / DropPath are the reference's;
    only the source of the Bernoulli draws is replaced: each module instance draws from the counter-based stream of
    oracle/rng_ref.py (site = 8 * block index + role, element index = position in the 2B-image batch the HIP path runs), which is
    what the HIP kernels regenerate.  The k-th call of a module tells the role: Tenc_x2 runs date 1 then date 2 (:666-670), Mlp.drop
    is called after the activation and after fc2 (:130,132), Block.drop_path for the attention and the Mlp branch (:246-247)."""
    out = {}
    c, B = 2, 2
    model = _changeformer_drop_model(c, B)
    x1 = sar_like("changeformer.drop.x1", (B, c, 224, 224))
    x2 = sar_like("changeformer.drop.x2", (B, c, 224, 224))
    lbl = seeded_labels("changeformer.drop.lbl", (B, 224, 224))
    crit = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)
    outs = model(x1, x2)
    loss = crit(outs[-1], lbl)
    loss.backward()
    out["seed_step"] = np.array([DROP_SEED, DROP_STEP])
    for i, o in enumerate(outs[:4]):
        out[f"train.out{i}"] = o.detach().numpy().copy()
    out["train.out4_sub"] = outs[4][:, :, ::8, ::8].detach().numpy().copy()
    out["train.loss"] = np.array(float(loss.detach()))
    for k, p in model.named_parameters():
        if p.grad is None:
            out[f"gstat.{k}"] = np.zeros(3)
            continue
        g = p.grad.detach().double()
        out[f"gstat.{k}"] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
        if k in CHANGEFORMER_GRAD_KEYS:
            out[f"grad.{k}"] = p.grad.detach().numpy().copy()
    print("changeformer drop train loss", float(loss.detach()))
    np.savez_compressed(os.path.join(OUT, "changeformer_drop.npz"), **out)


def gen_changeformer_bench():
    """BASELINE.json configs[3] as benchmarked, at batch 8: ChangeFormerV6(input_nc = 4: SLC tiles), 224 x 224, stochastic layers ON (on
    the counter-based stream, as gen_changeformer_drop), the synthetic SAR tiles of bench.py, ce+dice on the sigmoid map, on the REAL
    reference in fp32: the five outputs (sub-sampled), the loss, gradient statistics of every parameter."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    out = {}
    c, B = 4, 8
    (x1, x2), lbl = cd_inputs(make_batch(B, 224, 224, seed=1234, channels=c), ("pre_event_1", "post_event"))
    model = _changeformer_drop_model(c, B)
    crit = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)
    outs = model(x1, x2)
    loss = crit(outs[-1], lbl)
    loss.backward()
    out["seed_step"] = np.array([DROP_SEED, DROP_STEP])
    for i, o in enumerate(outs[:4]):
        out[f"train.out{i}"] = o.detach().numpy().copy()
    out["train.out4_sub"] = outs[4][:, :, ::8, ::8].detach().numpy().copy()
    out["train.argmax_sub"] = outs[4][::2].detach().argmax(1).numpy().astype(np.uint8)
    top2 = outs[4][::2].detach().topk(2, dim=1).values
    out["train.margin_sub"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    out["train.loss"] = np.array(float(loss.detach()))
    for k, p in model.named_parameters():
        if p.grad is None:
            out[f"gstat.{k}"] = np.zeros(3)
            continue
        g = p.grad.detach().double()
        out[f"gstat.{k}"] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
        if k in CHANGEFORMER_GRAD_KEYS:
            out[f"grad.{k}"] = p.grad.detach().numpy().copy()
    print("changeformer bench train loss", float(loss.detach()))
    np.savez_compressed(os.path.join(OUT, "changeformer_bench.npz"), **out)


def gen_changeformer_bench32_eval():
    """BASELINE.json configs[3] at its STATED batch (32): the GPU plan of that size is held to vectors of the real reference through
    eval mode, where every sample is independent of the rest of the batch (BatchNorm running statistics, stochastic layers off): the
    reference runs the first 8 of the 32 benchmark tiles (make_batch(32, seed 1234, 4 SLC bands)), fp32."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    ChangeFormerV6 = _import_changeformer_reference()
    c, B, n = 4, 32, 8
    (x1, x2), _ = cd_inputs(make_batch(B, 224, 224, seed=1234, channels=c), ("pre_event_1", "post_event"))
    model = ChangeFormerV6(input_nc=c, output_nc=3, decoder_softmax=True, embed_dim=256)
    seeded_fill_(model.state_dict())
    model.eval()
    out = {}
    with torch.no_grad():
        outs = model(x1[:n], x2[:n])
    for i, o in enumerate(outs[:4]):
        out[f"eval.out{i}"] = o.numpy().copy()
    out["eval.out4_sub"] = outs[4][:, :, ::8, ::8].numpy().copy()
    out["eval.argmax_sub"] = outs[4][::2].argmax(1).numpy().astype(np.uint8)
    top2 = outs[4][::2].topk(2, dim=1).values
    out["eval.margin_sub"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    print("changeformer bench32 eval: out4 mean", float(outs[4].mean()))
    np.savez_compressed(os.path.join(OUT, "changeformer_bench32_eval.npz"), **out)


def gen_snunet_dem_shard():
    """BASELINE.json configs[2] per-GPU shard: SNUNet-ECAM with 3 channels per date (VV, VH, DEM), batch 8 (global 64 over 8 ranks),
    224 x 224, on the REAL reference in fp32: train-mode logits, ce+dice loss, gradient statistics.  The DEM plane is a seeded smooth
    field shared by both dates (the trainer's torch.cat((image, dem), 1), change_detection_trainer.py:117-133)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from kurosiwo_amd.synthetic import cd_inputs, make_batch
    out = {}
    c, bc, B = 3, 32, 8
    (xA, xB), lbl = cd_inputs(make_batch(B, 224, 224, seed=4321), ("pre_event_1", "post_event"))
    dem = torch.nn.functional.interpolate(seeded_tensor("snunet_dem_shard.dem", (B, 1, 14, 14)), size=(224, 224), mode="bilinear", align_corners=False)
    xA, xB = torch.cat((xA, dem), 1), torch.cat((xB, dem), 1)
    model = _ref_model(c, bc)
    model.train()
    crit = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)
    logits = model(xA, xB)
    loss = crit(logits, lbl)
    loss.backward()
    out["train_logits_sub"] = logits[:, :, ::8, ::8].detach().numpy().copy()
    out["train_logits_absmax"] = np.array(float(logits.detach().abs().max()))
    out["train_argmax_sub"] = logits[::2].detach().argmax(1).numpy().astype(np.uint8)
    top2 = logits[::2].detach().topk(2, dim=1).values
    out["train_margin_sub"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    out["train_loss"] = np.array(float(loss))
    stats, _ = _grad_stats(model)
    for k, v in stats.items():
        out[f"gstat.{k}"] = v
    for k in ("conv_final.weight", "conv0_0.conv1.weight", "conv0_4.conv1.weight", "Up1_3.up.weight"):
        out[f"grad.{k}"] = dict(model.named_parameters())[k].grad.numpy().copy()
    print("snunet_dem_shard loss", float(loss))
    np.savez_compressed(os.path.join(OUT, "snunet_dem_shard.npz"), **out)


def gen_fcsiam():
    """FC-Siam-conc / FC-Siam-diff (N2 row): the REFERENCE modules (models/siam_conc.py, siam_diff.py import with torch alone).  Eval
    outputs, and one train-mode step with every nn.Dropout2d(p=0.2) ON, its plane mask drawn from the counter-based stream of
    oracle/rng_ref.py: site = 2 * (index of the layer among the BatchNorm'ed layers) + date, where the date is the call count of the
    module within the forward (the shared encoder modules run for date 1, then date 2: siam_conc.py:100-146)."""
    from models.siam_conc import SiamUnet_conc
    from models.siam_diff import SiamUnet_diff
    from oracle import fcsiam_ref as R
    from oracle import rng_ref as G
    c, B, S = 2, 2, 96
    for tag, cls in (("conc", SiamUnet_conc), ("diff", SiamUnet_diff)):
        out = {}
        model = cls(c, 3)
        seeded_fill_(model.state_dict())
        sd = model.state_dict()
        out["state_dict_keys"] = np.array(list(sd.keys()))
        out["state_dict_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
        x1 = sar_like(f"fcsiam.{tag}.eval.x1", (1, c, S, S))
        x2 = sar_like(f"fcsiam.{tag}.eval.x2", (1, c, S, S))
        model.eval()
        with torch.no_grad():
            out["eval.out"] = model(x1, x2).numpy().copy()
        model.train()
        for name, mod in model.named_modules():
            if isinstance(mod, torch.nn.Dropout2d):
                assert name.startswith("do") and abs(mod.p - 0.2) < 1e-12
                layer, calls = R.LAYERS.index(name[2:]), {"n": 0}

                def fwd(x, layer=layer, calls=calls):
                    site = 2 * layer + calls["n"]
                    calls["n"] += 1
                    m = G.scale_mask(DROP_SEED, DROP_STEP, site, 0.2, 0, (x.shape[0], x.shape[1]))
                    return x * torch.from_numpy(m)[:, :, None, None]
                mod.forward = fwd
        x1 = sar_like(f"fcsiam.{tag}.train.x1", (B, c, S, S))
        x2 = sar_like(f"fcsiam.{tag}.train.x2", (B, c, S, S))
        lbl = seeded_labels(f"fcsiam.{tag}.train.lbl", (B, S, S))
        crit = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)
        o = model(x1, x2)
        loss = crit(o, lbl)                                      # change_detection_trainer.py:138-166: the model output goes to the criterion
        loss.backward()
        out["seed_step"] = np.array([DROP_SEED, DROP_STEP])
        out["train.out"] = o.detach().numpy().copy()
        out["train.loss"] = np.array(float(loss.detach()))
        for k, p in model.named_parameters():
            g = p.grad.detach().double()
            out[f"gstat.{k}"] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
            if k.startswith(("conv11.", "conv21.", "upconv1.", "conv22d.", "conv12d.", "conv11d.", "bn21.", "bn32d.")):
                out[f"grad.{k}"] = p.grad.detach().numpy().copy()
        sd = model.state_dict()
        for k in ("bn11", "bn43", "bn43d", "bn12d"):
            out[f"bn.{k}.running_mean"] = sd[f"{k}.running_mean"].numpy().copy()
            out[f"bn.{k}.running_var"] = sd[f"{k}.running_var"].numpy().copy()
            out[f"bn.{k}.num_batches_tracked"] = sd[f"{k}.num_batches_tracked"].numpy().copy()
        print("fcsiam", tag, "train loss", float(loss.detach()), "keys", len(sd))
        np.savez_compressed(os.path.join(OUT, f"fcsiam_{tag}.npz"), **out)


def gen_bitcd():
    """BIT-CD as shipped (configs/method/bit-cd/bit_cd.json: net_G = base_resnet18): the REFERENCE `define_G` network (models/bit_cd.py imports
    with torch + einops alone), seeded weights, eval output and one train-mode step (loss, gradient statistics, BatchNorm running stats)."""
    from models.bit_cd import define_G
    c, B, S = 2, 2, 64
    out = {}
    model = define_G({"net_G": "base_resnet18", "init_type": "normal", "init_gain": 0.02}, c)
    seeded_fill_(model.state_dict())
    sd = model.state_dict()
    out["state_dict_keys"] = np.array(list(sd.keys()))
    out["state_dict_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    x1 = sar_like("bitcd.eval.x1", (1, c, S, S))
    x2 = sar_like("bitcd.eval.x2", (1, c, S, S))
    model.eval()
    with torch.no_grad():
        out["eval.out"] = model(x1, x2).numpy().copy()
    model.train()
    x1 = sar_like("bitcd.train.x1", (B, c, S, S))
    x2 = sar_like("bitcd.train.x2", (B, c, S, S))
    lbl = seeded_labels("bitcd.train.lbl", (B, S, S))
    crit = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)
    o = model(x1, x2)
    loss = crit(o, lbl)
    loss.backward()
    out["train.out"] = o.detach().numpy().copy()
    out["train.loss"] = np.array(float(loss.detach()))
    for k, p in model.named_parameters():
        if p.grad is None:
            out[f"gstat.{k}"] = np.zeros(3)
            continue
        g = p.grad.detach().double()
        out[f"gstat.{k}"] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
        if k in ("resnet.conv1.weight", "resnet.layer2.0.downsample.0.weight", "resnet.bn1.weight", "classifier.0.weight", "classifier.3.weight",
                 "classifier.3.bias", "conv_pred.bias", "resnet.layer3.0.bn2.bias"):
            out[f"grad.{k}"] = p.grad.detach().numpy().copy()
    sd = model.state_dict()
    for k in ("resnet.bn1", "resnet.layer2.0.downsample.1", "resnet.layer4.1.bn2", "classifier.1"):
        out[f"bn.{k}.running_mean"] = sd[f"{k}.running_mean"].numpy().copy()
        out[f"bn.{k}.running_var"] = sd[f"{k}.running_var"].numpy().copy()
        out[f"bn.{k}.num_batches_tracked"] = sd[f"{k}.num_batches_tracked"].numpy().copy()
    print("bitcd train loss", float(loss.detach()), "keys", len(sd))
    np.savez_compressed(os.path.join(OUT, "bitcd.npz"), **out)


def gen_bitcd_transformer():
    """The three BASE_Transformer variants of the REFERENCE define_G (models/bit_cd.py:690-700, 802-934): seeded weights, eval output,
    one train-mode step (loss, gradient statistics + the full gradients of the token-path parameters, BatchNorm running statistics)."""
    from models.bit_cd import define_G
    c, B, S = 2, 2, 64
    for net_G in ("base_transformer_pos_s4", "base_transformer_pos_s4_dd8", "base_transformer_pos_s4_dd8_dedim8"):
        out = {}
        model = define_G({"net_G": net_G, "init_type": "normal", "init_gain": 0.02}, c)
        seeded_fill_(model.state_dict())
        sd = model.state_dict()
        out["state_dict_keys"] = np.array(list(sd.keys()))
        out["state_dict_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
        x1 = sar_like("bitcd.eval.x1", (1, c, S, S))
        x2 = sar_like("bitcd.eval.x2", (1, c, S, S))
        model.eval()
        with torch.no_grad():
            out["eval.out"] = model(x1, x2).numpy().copy()
            out["eval.tokens"] = model.tokens.numpy().copy()                  # the encoder's output (BASE_Transformer keeps it, :919)
        model.train()
        x1 = sar_like("bitcd.train.x1", (B, c, S, S))
        x2 = sar_like("bitcd.train.x2", (B, c, S, S))
        lbl = seeded_labels("bitcd.train.lbl", (B, S, S))
        crit = BCEandDiceLoss(weights=CLASS_WEIGHTS, ignore_index=3, use_softmax=True)
        o = model(x1, x2)
        loss = crit(o, lbl)
        loss.backward()
        out["train.out"] = o.detach().numpy().copy()
        out["train.loss"] = np.array(float(loss.detach()))
        for k, p in model.named_parameters():
            if p.grad is None:
                out[f"gstat.{k}"] = np.zeros(3)
                continue
            g = p.grad.detach().double()
            out[f"gstat.{k}"] = np.array([float(g.norm()), float(g.sum()), float(g.abs().max())])
            if k.startswith(("pos_embedding", "conv_a", "transformer.", "transformer_decoder.layers.0.", "conv_pred")) or k in (
                    "resnet.conv1.weight", "classifier.3.bias", "resnet.layer3.0.bn2.bias"):
                out[f"grad.{k}"] = p.grad.detach().numpy().copy()
        sd = model.state_dict()
        for k in ("resnet.bn1", "resnet.layer3.1.bn2", "resnet.layer4.1.bn2", "classifier.1"):
            out[f"bn.{k}.running_mean"] = sd[f"{k}.running_mean"].numpy().copy()
            out[f"bn.{k}.running_var"] = sd[f"{k}.running_var"].numpy().copy()
            out[f"bn.{k}.num_batches_tracked"] = sd[f"{k}.num_batches_tracked"].numpy().copy()
        print(net_G, "train loss", float(loss.detach()), "keys", len(sd))
        np.savez_compressed(os.path.join(OUT, f"bitcd_{net_G}.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    if not only or "loss" in only:
        gen_loss()
    if not only or "snunet" in only:
        gen_snunet_small()
        gen_snunet_full()
    if not only or "snunet_bench" in only:
        gen_snunet_bench()
    if not only or "floodvit" in only:
        gen_floodvit("small", FLOODVIT_SMALL, 2)
        gen_floodvit("full", FLOODVIT_FULL, 1)
    if not only or "floodvit_heads" in only:
        gen_floodvit("small_mlp", FLOODVIT_SMALL, 1, head="mlp")
        gen_floodvit("small_linear", FLOODVIT_SMALL, 1, head="linear")
    if not only or "changeformer" in only:
        gen_changeformer()
    if not only or "changeformer_slc" in only:
        gen_changeformer_slc()
    if not only or "changeformer_drop" in only:
        gen_changeformer_drop()
    if not only or "changeformer_bench" in only:
        gen_changeformer_bench()
    if not only or "floodvit_bench" in only:
        gen_floodvit_bench()
    if not only or "changeformer_bench32_eval" in only:
        gen_changeformer_bench32_eval()
    if not only or "snunet_dem_shard" in only:
        gen_snunet_dem_shard()
    if not only or "fcsiam" in only:
        gen_fcsiam()
    if not only or "bitcd" in only:
        gen_bitcd()
    if not only or "bitcd_transformer" in only:
        gen_bitcd_transformer()
    if not only or "mae" in only:
        gen_mae()
