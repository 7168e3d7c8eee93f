"""Oracle (TEST INFRASTRUCTURE) for row M1 of SURVEY.md §8(a): argmax ->
4x4 confusion matrix -> per-class accuracy / F1 / precision / recall / IoU.

Follows /root/reference/utilities/utilities.py:228-265 (torchmetrics multiclass,
num_classes = 3 + 1, average='none', multidim_average='global', ignore_index=3)
and the call sites /root/reference/training/change_detection_trainer.py:152,
184-189 (``predictions = output.argmax(1)``; ``mIoU = iou[:3].mean()``).

PARITY UNPINNED BY IMPORT: torchmetrics (0.11.4, requirements.txt:3) is a
third-party dependency absent from /root/reference and from this image; the
formulas below restate its documented multiclass stat-scores semantics
(target == ignore_index pixels are dropped; 0/0 -> 0) and are pinned by
hand-computed cases in tests/test_host_cpu.py (test_metrics_*; the per-group
kernel is held to these functions in tests/test_gpu_eval.py).  Pure numpy integers.
"""
import numpy as np

NUM_CLASSES = 4
IGNORE_INDEX = 3


def argmax_lowest_index(logits):
    """torch.argmax tie semantics used here: lowest index wins."""
    return np.argmax(np.asarray(logits), axis=1).astype(np.int64)


def confusion_matrix(pred, target, num_classes=NUM_CLASSES, ignore_index=IGNORE_INDEX):
    """CM[t, p] = #pixels with target t and prediction p, target != ignore_index."""
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    target = np.asarray(target).reshape(-1).astype(np.int64)
    keep = target != ignore_index
    idx = target[keep] * num_classes + pred[keep]
    return np.bincount(idx, minlength=num_classes * num_classes).reshape(num_classes, num_classes).astype(np.int64)


def _safe_div(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return np.where(b > 0, a / np.where(b > 0, b, 1), 0.0)


def metrics_from_cm(cm):
    cm = np.asarray(cm, dtype=np.int64)
    tp = np.diag(cm)
    row = cm.sum(axis=1)       # support per target class (tp + fn)
    col = cm.sum(axis=0)       # predicted per class (tp + fp)
    recall = _safe_div(tp, row)
    precision = _safe_div(tp, col)
    f1 = _safe_div(2 * tp, row + col)
    iou = _safe_div(tp, row + col - tp)
    return {
        "accuracy": recall,            # multiclass per-class accuracy == recall
        "recall": recall,
        "precision": precision,
        "f1": f1,
        "iou": iou,
        "miou": iou[:3].mean(),        # change_detection_trainer.py:189
    }
