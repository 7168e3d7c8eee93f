"""Loss surface of the reference on the fused HIP kernel (rows L1/L2).

  create_loss(configs, mode)      <- /root/reference/utilities/utilities.py:307-347
  BCEandDiceLoss(weights, ignore_index, use_softmax)
                                  <- /root/reference/utilities/bce_and_dice.py:7-24
                                     (+ utilities/dice.py:93-137)
  CrossEntropyLoss(weight, ignore_index)  == nn.CrossEntropyLoss as used by create_loss

callable(preds fp32 [B,3,H,W], lbl int64 [B,H,W]) -> 0-dim tensor with autograd.
"""
import torch
import torch.nn as nn

from . import _lib
from .runtime import require_gpu, stream_ptr


class _CEDiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, cw, with_dice, ignore_index):
        lib = _lib.load()
        B, Cc, H, W = logits.shape
        out3 = torch.empty(3, dtype=torch.float32, device=logits.device)
        ws = torch.empty(lib.ksmi_loss_workspace(B, H * W), dtype=torch.uint8, device=logits.device)
        _lib.check(lib.ksmi_ce_dice_forward(logits.data_ptr(), labels.data_ptr(), cw.data_ptr(), int(with_dice),
                                            out3.data_ptr(), ws.data_ptr(), B, H * W, ignore_index, stream_ptr()), "ce_dice_forward")
        ctx.save_for_backward(logits, labels, cw, ws)
        ctx.with_dice, ctx.ignore_index = with_dice, ignore_index
        ctx.parts = out3
        return out3[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        logits, labels, cw, ws = ctx.saved_tensors
        lib = _lib.load()
        B, Cc, H, W = logits.shape
        dl = torch.empty_like(logits)
        gs = grad_out.contiguous().float()
        _lib.check(lib.ksmi_ce_dice_backward(logits.data_ptr(), labels.data_ptr(), cw.data_ptr(), int(ctx.with_dice),
                                             ws.data_ptr(), gs.data_ptr(), dl.data_ptr(), B, H * W, ctx.ignore_index,
                                             stream_ptr()), "ce_dice_backward")
        return dl, None, None, None, None


class _HipLoss(nn.Module):
    def __init__(self, weights, ignore_index, with_dice):
        super().__init__()
        w = torch.as_tensor(weights if weights is not None else [1.0, 1.0, 1.0], dtype=torch.float32)
        if w.numel() != 3:
            raise _lib.KsmiError("the HIP loss supports num_classes == 3 (reference configs/config.json:13)")
        self.register_buffer("weight", w)
        self.ignore_index = -100 if ignore_index is None else int(ignore_index)
        self.with_dice = with_dice
        self.last_parts = None

    def forward(self, preds, lbl):
        require_gpu(preds)
        if preds.dim() != 4 or preds.shape[1] != 3:
            raise ValueError(f"Invalid input shape, we expect Bx3xHxW. Got: {tuple(preds.shape)}")
        if preds.shape[-2:] != lbl.shape[-2:]:
            raise ValueError(f"input and target shapes must be the same. Got: {tuple(preds.shape)} {tuple(lbl.shape)}")
        if lbl.dtype != torch.int64:
            raise ValueError(f"labels must be torch.int64. Got: {lbl.dtype}")
        if self.weight.device != preds.device:
            self.weight = self.weight.to(preds.device)
        return _CEDiceFn.apply(preds.contiguous().float(), lbl.contiguous(), self.weight, self.with_dice, self.ignore_index)


class BCEandDiceLoss(_HipLoss):
    """softmax-CE (weighted, ignore_index) + softmax-Dice; `use_softmax` must be True as in create_loss."""

    def __init__(self, weights=None, ignore_index=None, use_softmax=False):
        if not use_softmax:
            raise _lib.KsmiError("BCEandDiceLoss(HIP): only use_softmax=True (the reference's create_loss setting) is implemented")
        super().__init__(weights, ignore_index, True)


class CrossEntropyLoss(_HipLoss):
    def __init__(self, weight=None, ignore_index=-100):
        super().__init__(weight, ignore_index, False)


def create_loss(configs, mode="val"):
    lf = configs["loss_function"]
    cw = configs.get("class_weights", [1.0, 1.0, 1.0])
    dev = configs.get("device", "cuda")
    if lf == "cross_entropy":
        if mode == "train":
            print("Creating cross entropy loss with class weights")
            print(torch.tensor(cw))
            return CrossEntropyLoss(weight=cw, ignore_index=3).to(dev)
        return CrossEntropyLoss(ignore_index=3).to(dev)
    if lf == "ce+dice":
        return BCEandDiceLoss(weights=cw, ignore_index=3, use_softmax=True).to(dev)
    raise NotImplementedError(f"loss_function={lf!r}: only 'cross_entropy' and 'ce+dice' have HIP kernels "
                              "(iou/dice are third-party smp losses, focal needs torch.hub/network)")
