"""Evaluation metric of the flow path: ``cal_epe`` (reference models/utils/flow.py:64-88).

Not a kernel: a handful of reductions over the final flow field, run once per evaluation
batch with torch ops on whatever device the flows live on.  Restated as-is, including the
reference's quirk in the 'mean' branch (the errors of the *valid* pixels are overwritten with
1e8 before the '<t>px' ratios are taken, flow.py:79) -- pass ``fix_threshold_quirk=True`` for
the evidently intended behaviour.
"""
from __future__ import annotations

import torch

__all__ = ['cal_epe']


def cal_epe(flow_tgt: torch.Tensor, flow_pred: torch.Tensor, mask, max_flow: float = 400,
            reduction: str = 'mean', threshs=(1, 3, 5), fix_threshold_quirk: bool = False):
    mag = torch.sum(flow_tgt ** 2, dim=1).sqrt()
    valid = (mag < max_flow) & (mask >= 0.5) if mask is not None else (mag < max_flow)
    err = torch.sum((flow_tgt - flow_pred) ** 2, dim=1).sqrt()
    if reduction == 'none':
        return err * valid.to(err)
    acc = {}
    if reduction == 'mean':
        total = valid.sum(dim=(-1, -2)) + 1e-10
        acc['mean'] = (err * valid.to(err)).sum(dim=(-1, -2)) / total
        thr_err = torch.where(valid, err, torch.full_like(err, 1e8)) if fix_threshold_quirk \
            else torch.where(valid, torch.full_like(err, 1e8), err)
        for t in threshs:
            acc[f'{t}px'] = (thr_err < t).sum(dim=(-1, -2)) / total
    elif reduction == 'total_mean':
        total = valid.sum(dim=(-1, -2, -3)) + 1e-10
        acc['mean'] = (err * valid.to(err.dtype)).sum(dim=(-1, -2, -3)) / total
        for t in threshs:
            acc[f'{t}px'] = (err[valid] < t).sum() / total
    else:
        raise ValueError(reduction)
    return acc
