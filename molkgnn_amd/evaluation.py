"""Evaluation metrics of the reference (``evaluation.py:11-127``) on torch tensors.

Same function names and return values as the reference's numpy / scikit-learn code, computed with torch sorts and
cumulative sums in float64 on whatever device the predictions live on (the validation set's scores never have to
leave the GPU).  ``roc_curve`` is restated with scikit-learn's semantics -- one threshold per distinct score, the
``drop_intermediate`` thinning of collinear points (which matters for logAUC: the trapezoids are taken in
log10(FPR)), the leading (0, 0) point -- and ``np.interp``'s convention for repeated abscissae.
Pinned by ``tests/golden/g8_metrics.npz`` (numbers produced by the reference's own functions).
"""
from __future__ import annotations

import math
from typing import Tuple

import torch


def _as_vectors(true_y, predicted_score) -> Tuple[torch.Tensor, torch.Tensor]:
    y = torch.as_tensor(true_y).reshape(-1)
    s = torch.as_tensor(predicted_score).reshape(-1).to(torch.float64)
    y = y.to(s.device)
    if y.numel() != s.numel():
        raise ValueError("true_y and predicted_score differ in length")
    return (y == 1).to(torch.float64), s


def roc_curve(true_y, predicted_score, drop_intermediate: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """(fpr, tpr) as ``sklearn.metrics.roc_curve(true_y, predicted_score, pos_label=1)`` returns them."""
    y, s = _as_vectors(true_y, predicted_score)
    order = torch.sort(s, descending=True, stable=True).indices
    s, y = s[order], y[order]
    n = s.numel()
    distinct = torch.nonzero(s[1:] != s[:-1]).reshape(-1)
    idx = torch.cat([distinct, torch.tensor([n - 1], device=s.device)])
    tps = torch.cumsum(y, 0)[idx]
    fps = 1.0 + idx.to(torch.float64) - tps
    if drop_intermediate and fps.numel() > 2:
        d2f = fps[2:] - 2 * fps[1:-1] + fps[:-2]
        d2t = tps[2:] - 2 * tps[1:-1] + tps[:-2]
        keep = torch.cat([torch.tensor([True], device=s.device), (d2f != 0) | (d2t != 0), torch.tensor([True], device=s.device)])
        fps, tps = fps[keep], tps[keep]
    zero = torch.zeros(1, dtype=torch.float64, device=s.device)
    fps, tps = torch.cat([zero, fps]), torch.cat([zero, tps])
    fpr = fps / fps[-1] if float(fps[-1]) > 0 else torch.full_like(fps, float("nan"))
    tpr = tps / tps[-1] if float(tps[-1]) > 0 else torch.full_like(tps, float("nan"))
    return fpr, tpr


def _interp(xq: torch.Tensor, xp: torch.Tensor, fp: torch.Tensor) -> torch.Tensor:
    """``np.interp`` for non-decreasing ``xp`` (a query equal to repeated abscissae takes the last of them)."""
    j = torch.searchsorted(xp, xq, right=True) - 1
    j = j.clamp(0, xp.numel() - 1)
    j1 = (j + 1).clamp(max=xp.numel() - 1)
    x0, x1, y0, y1 = xp[j], xp[j1], fp[j], fp[j1]
    w = torch.where(x1 > x0, (xq - x0) / (x1 - x0), torch.zeros_like(xq))
    out = y0 + w * (y1 - y0)
    out = torch.where(xq <= xp[0], fp[0].expand_as(out), out)
    out = torch.where(xq >= xp[-1], fp[-1].expand_as(out), out)
    return out


def _trapz(y: torch.Tensor, x: torch.Tensor) -> float:
    return float(((x[1:] - x[:-1]) * (y[1:] + y[:-1]) * 0.5).sum())


def calculate_logAUC(true_y, predicted_score, FPR_range=(0.001, 0.1)) -> float:
    """Area under the ROC curve over ``log10(FPR)`` in ``FPR_range``, normalised by the range (evaluation.py:11-79)."""
    if FPR_range is None:
        raise Exception('FPR range cannot be None')
    lower, upper = FPR_range
    if lower >= upper:
        raise Exception('FPR upper_bound must be greater than lower_bound')
    fpr, tpr = roc_curve(true_y, predicted_score)
    q = torch.tensor([lower, upper], dtype=torch.float64, device=fpr.device)
    tpr = torch.sort(torch.cat([tpr, _interp(q, fpr, tpr)])).values
    fpr = torch.sort(torch.cat([fpr, q])).values
    x = torch.log10(fpr)
    lo, hi = math.log10(lower), math.log10(upper)
    lo_t = torch.log10(torch.tensor(lower, dtype=torch.float64, device=fpr.device))
    hi_t = torch.log10(torch.tensor(upper, dtype=torch.float64, device=fpr.device))
    i0 = int(torch.nonzero(x == lo_t).reshape(-1)[-1])
    i1 = int(torch.nonzero(x == hi_t).reshape(-1)[-1])
    return _trapz(tpr[i0:i1 + 1], x[i0:i1 + 1]) / (hi - lo)


def calculate_auc(true_y, predicted_score) -> float:
    """``roc_auc_score``, or ``-1`` when it is undefined (reference ``evaluation.py:81-86``: ``try: roc_auc_score(...)
    except: -1``).  With only one class present the reference's pinned scikit-learn (``requirements.txt:101``: 1.0.2) raises
    ``ValueError`` and the caller gets ``-1``; the build container's scikit-learn 1.7 warns and returns NaN instead (which
    is what the golden file, generated here, holds) -- the pinned behaviour is the contract."""
    y, _ = _as_vectors(true_y, predicted_score)
    if float(y.sum()) == 0.0 or float(y.sum()) == float(y.numel()):
        return -1
    fpr, tpr = roc_curve(true_y, predicted_score)
    return _trapz(tpr, fpr)


def _confusion(true_y, predicted_score, cutoff: float):
    y, s = _as_vectors(true_y, predicted_score)
    pred = (torch.sigmoid(s) > cutoff).to(torch.float64)
    tp = float((pred * y).sum())
    fp = float((pred * (1 - y)).sum())
    fn = float(((1 - pred) * y).sum())
    tn = float(((1 - pred) * (1 - y)).sum())
    return tn, fp, fn, tp


def calculate_ppv(true_y, predicted_score, cutoff: float = 0.5) -> float:
    tn, fp, fn, tp = _confusion(true_y, predicted_score, cutoff)
    return tp / (tp + fp) if (tp + fp) != 0 else float("nan")


def calculate_accuracy(true_y, predicted_score) -> float:
    tn, fp, fn, tp = _confusion(true_y, predicted_score, 0.5)
    tot = tp + fp + tn + fn
    return (tp + tn) / tot if tot != 0 else float("nan")


def calculate_f1_score(true_y, predicted_score) -> float:
    tn, fp, fn, tp = _confusion(true_y, predicted_score, 0.5)
    den = 2 * tp + fp + fn
    return 2 * tp / den if den != 0 else 0.0
