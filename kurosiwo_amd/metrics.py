"""Metrics of the reference trainers from ONE fused argmax -> 4x4 confusion-matrix kernel
(row M1): utilities/utilities.py:228-265 and
training/change_detection_trainer.py:152,184-189.
"""
import torch

from . import _lib
from .runtime import require_gpu, stream_ptr


class ConfusionMetrics:
    """Accumulates CM[target, pred] (int64, on device) over batches; compute() derives the five
    torchmetrics quantities (per class, 4 classes, target==3 ignored, 0/0 -> 0)."""

    def __init__(self, device, num_classes=4, ignore_index=3):
        self.cm = torch.zeros((4, 4), dtype=torch.int64, device=device)
        self.ignore_index = ignore_index
        self.num_classes = num_classes

    def reset(self):
        self.cm.zero_()

    def update(self, logits, target, return_predictions=False):
        require_gpu(logits)
        B, Cc, H, W = logits.shape
        logits = logits.contiguous().float()
        if target.dtype != torch.int64:          # the kernel reads int64 labels (the reference's masks are .long(), Dataset.py:824-860)
            target = target.long()
        pred = torch.empty((B, H, W), dtype=torch.int64, device=logits.device) if return_predictions else None
        lib = _lib.load()
        _lib.check(lib.ksmi_argmax_confusion(logits.data_ptr(), target.contiguous().data_ptr(),
                                             pred.data_ptr() if pred is not None else None, self.cm.data_ptr(),
                                             B, Cc, H * W, self.ignore_index, stream_ptr()), "argmax_confusion")
        return pred

    def compute(self):
        return metrics_from_cm(self.cm)


def metrics_from_cm(cm):
    cm = cm.to(torch.float64).cpu()
    tp = cm.diag()
    row, col = cm.sum(1), cm.sum(0)

    def sdiv(a, b):
        return torch.where(b > 0, a / torch.where(b > 0, b, torch.ones_like(b)), torch.zeros_like(a))
    recall = sdiv(tp, row)
    precision = sdiv(tp, col)
    f1 = sdiv(2 * tp, row + col)
    iou = sdiv(tp, row + col - tp)
    return {"accuracy": recall, "recall": recall, "precision": precision, "f1": f1, "iou": iou,
            "miou": iou[:3].mean()}
