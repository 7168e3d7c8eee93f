"""Oracle (TEST INFRASTRUCTURE) for row L1/L2 of SURVEY.md §8(a): the
"BCE+Dice" loss, which in the reference is softmax-CrossEntropy + softmax-Dice.

Follows
  * /root/reference/utilities/bce_and_dice.py:18-24  (total = dice + ce)
  * /root/reference/utilities/dice.py:14-59          (one_hot(+eps))
  * /root/reference/utilities/dice.py:93-137         (DiceLoss.forward)
  * /root/reference/utilities/utilities.py:307-347   (create_loss dispatch)

Written as explicit float64 numpy arithmetic (closed forms, including the
analytic gradient) so that it is an *independent* check of the HIP kernel and of
torch autograd, not a re-run of the same library calls.  Pinned by KAT-loss-1
 and tests/golden/loss_*.npz.
"""
import numpy as np

IGNORE_INDEX = 3          # utilities/utilities.py:316,346
DICE_EPS = 1e-6           # dice.py:91 (self.eps) and dice.py:18 (one_hot eps)


def _softmax(x):
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=1, keepdims=True)


def ce_dice_forward(logits, labels, weights=(1.0, 1.0, 1.0), ignore_index=IGNORE_INDEX,
                    with_dice=True, with_grad=False):
    """logits [B,C,H,W] float, labels [B,H,W] int in {0..C-1, ignore_index}.

    Returns dict(total, ce, dice[, grad]) in float64.
    """
    x = np.asarray(logits, dtype=np.float64)
    t = np.asarray(labels).astype(np.int64)
    B, C, H, W = x.shape
    w = np.asarray(weights, dtype=np.float64)
    p = _softmax(x)
    valid = t != ignore_index
    t0 = np.where(valid, t, 0)                      # dice.py:115-119 (ignored -> class 0)
    onehot_idx = np.eye(C, dtype=np.float64)[t0].transpose(0, 3, 1, 2)   # [B,C,H,W]

    # --- CrossEntropyLoss(weight=w, ignore_index) : weighted mean over valid pixels
    logp = np.log(p)
    nll = -(onehot_idx * logp).sum(axis=1)          # [B,H,W]
    wt = w[t0] * valid
    wsum = wt.sum()
    ce = (wt * nll).sum() / wsum

    out = {"ce": ce}
    grad = None
    if with_grad:
        # d ce / d x = wt * (p - onehot) / wsum
        grad = (wt[:, None] * (p - onehot_idx)) / wsum

    if with_dice:
        # one_hot is built as int64 then "+ eps" promotes to fp32 (dice.py:57-59): the
        # fp32 values are fl32(1 + 1e-6) and fl32(1e-6).
        one = np.float64(np.float32(1.0) + np.float32(DICE_EPS))
        zero = np.float64(np.float32(DICE_EPS))
        oh = np.where(onehot_idx > 0.5, one, zero)
        inter = (p * oh).sum(axis=(1, 2, 3))        # [B]
        card = (p + oh).sum(axis=(1, 2, 3))
        dice_score = 2.0 * inter / (card + DICE_EPS)
        dice = (1.0 - dice_score).mean()
        out["dice"] = dice
        out["total"] = dice + ce
        if with_grad:
            # d dice / d p_c = -(1/B) * ( 2*oh_c/(card+eps) - 2*I/(card+eps)^2 )
            a = (2.0 / (card + DICE_EPS))[:, None, None, None]
            b = (2.0 * inter / (card + DICE_EPS) ** 2)[:, None, None, None]
            g_p = -(a * oh - b) / B
            # softmax jacobian: dx_c = p_c * (g_c - sum_j g_j p_j)
            dot = (g_p * p).sum(axis=1, keepdims=True)
            grad = grad + p * (g_p - dot)
    else:
        out["total"] = ce
    if with_grad:
        out["grad"] = grad
    return out


def loss_from_config(configs, mode="val"):
    """Mirror of create_loss (utilities/utilities.py:307-347) for the two loss
    functions in scope.  Returns a callable(logits, labels, with_grad) -> dict."""
    lf = configs["loss_function"]
    cw = configs.get("class_weights", [1.0, 1.0, 1.0])
    if lf == "cross_entropy":
        w = cw if mode == "train" else [1.0, 1.0, 1.0]      # :314-321
        return lambda x, t, with_grad=False: ce_dice_forward(x, t, w, with_dice=False, with_grad=with_grad)
    if lf == "ce+dice":
        return lambda x, t, with_grad=False: ce_dice_forward(x, t, cw, with_dice=True, with_grad=with_grad)
    raise NotImplementedError(lf)
