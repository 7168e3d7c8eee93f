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


class GroupedConfusion:
    """The overall confusion matrix of an evaluation loop plus up to two families of per-group matrices (per activation id, per climate
    zone: training/change_detection_trainer.py:331-337, 437-472 keep one torchmetrics object per group and update it sample by sample),
    all accumulated by ONE launch per batch (ksmi_argmax_confusion_grouped): the per-sample group keys arrive with the batch on the host
    (Dataset.py: `clz`, `activ`), become int32 slot indices there, and ride to the device in one small asynchronous copy -- no
    per-sample launch, slice or host synchronisation.  `families` = [keys of family a, keys of family b]; groups[f][key] is a
    ConfusionMetrics whose .cm is a VIEW of the family's [G, 4, 4] table, so compute() / all-reduce code is unchanged."""

    def __init__(self, device, families=(), ignore_index=3):
        if len(families) > 2:
            raise _lib.KsmiError("GroupedConfusion: at most two group families per launch")
        self.total = ConfusionMetrics(device, ignore_index=ignore_index)
        self.ignore_index = ignore_index
        self.tables, self.index, self.groups = [], [], []
        for keys in families:
            keys = list(keys)
            t = torch.zeros((max(len(keys), 1), 4, 4), dtype=torch.int64, device=device)
            self.tables.append(t)
            self.index.append({k: i for i, k in enumerate(keys)})
            g = {}
            for i, k in enumerate(keys):
                cmx = ConfusionMetrics(device, ignore_index=ignore_index)
                cmx.cm = t[i]
                g[k] = cmx
            self.groups.append(g)

    def update(self, logits, target, keys=()):
        """keys[f] = the per-sample group key of family f (host sequence / CPU tensor of length B; unknown keys belong to no group)"""
        require_gpu(logits)
        B, Cc, H, W = logits.shape
        logits = logits.contiguous().float()
        if target.dtype != torch.int64:
            target = target.long()
        slots = []
        for f, table in enumerate(self.tables):
            if f < len(keys) and keys[f] is not None and self.index[f]:
                host = torch.tensor([self.index[f].get(int(k), -1) for k in keys[f]], dtype=torch.int32)
                if host.numel() != B:
                    raise _lib.KsmiError(f"GroupedConfusion: {host.numel()} group keys for a batch of {B}")
                slots.append((host.to(logits.device, non_blocking=True), table))
            else:
                slots.append((None, None))
        while len(slots) < 2:
            slots.append((None, None))
        (sa, ta), (sb, tb) = slots
        P = lambda t: t.data_ptr() if t is not None else None
        _lib.check(_lib.load().ksmi_argmax_confusion_grouped(logits.data_ptr(), target.contiguous().data_ptr(), None, self.total.cm.data_ptr(),
                                                             P(sa), P(ta), P(sb), P(tb), B, Cc, H * W, self.ignore_index, stream_ptr()),
                   "argmax_confusion_grouped")
        self._keep = (sa, sb, logits, target)      # (alive until the next update: the launch is asynchronous)


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
