"""Evaluation side of the flow path: ``cal_epe`` (reference
models/utils/flow.py:64-88) and the ground-truth flow it is measured against --
``get_flow_from_delta_pose_and_depth`` (models/utils/pose.py:92-121) and ``filter_flow_by_mask``
(models/utils/flow.py:6-26), as the RAFT refiners build it (raft_refiner_flow_mask.py:180-191).
All three run on HIP kernels: the dense re-projection of the hot path, one element-wise filter, and
``scf_cal_epe`` (metrics.hip: one pass over the two flow fields, fixed-order reductions).

``cal_epe`` is restated as-is, including the reference's quirk in the 'mean' branch (the errors of
the *valid* pixels are overwritten with 1e8 before the '<t>px' ratios are taken, flow.py:79) -- pass
``fix_threshold_quirk=True`` for the evidently intended behaviour.
"""
from __future__ import annotations

import torch

__all__ = ['cal_epe', 'get_flow_from_delta_pose_and_depth', 'filter_flow_by_mask',
           'eval_pose_error', 'eval_rot_error', 'eval_tran_error']


def get_flow_from_delta_pose_and_depth(rotation_src, translation_src, rotation_dst, translation_dst,
                                       depth_src, k, invalid_num: float = 400):
    """models/utils/pose.py:92-121, same argument order -> (N,2,H,W); the reference's per-sample
    nonzero / mm / scatter loop is one launch of ``reproject_flow_kernel``."""
    from . import ops
    return ops.reproject_flow(depth_src.contiguous(), k.contiguous(), rotation_src.contiguous(),
                              translation_src.contiguous(), rotation_dst.contiguous(),
                              translation_dst.contiguous(), invalid_num=float(invalid_num))


def filter_flow_by_mask(flow, gt_mask, invalid_num: float = 400, mode: str = 'bilinear',
                        align_corners: bool = False):
    """models/utils/flow.py:6-26: modifies ``flow`` in place and returns it, like the reference."""
    if mode != 'bilinear':
        raise NotImplementedError("filter_flow_by_mask: mode='bilinear'")
    from . import ops
    if not flow.is_contiguous():
        raise ValueError('filter_flow_by_mask works in place on a contiguous flow tensor')
    return ops.filter_flow_by_mask_(flow, gt_mask.to(torch.float32).contiguous(), invalid_num, align_corners)


def cal_epe(flow_tgt: torch.Tensor, flow_pred: torch.Tensor, mask, max_flow: float = 400,
            reduction: str = 'mean', threshs=(1, 3, 5), fix_threshold_quirk: bool = False):
    """``cal_epe`` (models/utils/flow.py:64-88), same arguments and returns (a dict of tensors, or the
    masked error map for ``reduction='none'``): one HIP pass over the two flow fields
    (``scf_cal_epe``: |delta|, validity, masked sums and the threshold counts, fixed-order
    reductions) plus a one-block combine.  GPU tensors only, like every other operator here."""
    import ctypes as C
    from . import _lib, ops
    if reduction not in ('none', 'mean', 'total_mean'):
        raise ValueError(reduction)
    if flow_tgt.dim() != 4 or flow_tgt.shape[1] != 2 or flow_tgt.shape != flow_pred.shape:
        raise _lib.ScflowHipError('cal_epe: flow_tgt / flow_pred must be equal-shape (N,2,H,W)')
    # any float dtype / memory layout, like the reference's torch expressions (the arithmetic is fp32, as it is
    # there for fp32 inputs); the device must be the GPU: there is no CPU path in this package
    flow_tgt = flow_tgt.to(torch.float32).contiguous()
    flow_pred = flow_pred.to(torch.float32).contiguous()
    pt, pp = ops._dense(flow_tgt, 'flow_tgt'), ops._dense(flow_pred, 'flow_pred')
    n, _, h, w = flow_tgt.shape
    pm = None
    if mask is not None:
        mask = mask.to(torch.float32).contiguous()
        if tuple(mask.shape) != (n, h, w):
            raise _lib.ScflowHipError(f'cal_epe: mask has shape {tuple(mask.shape)}, expected {(n, h, w)}')
        pm = ops._dense(mask, 'mask')
    lib = _lib.load()
    dev = flow_tgt.device
    threshs = tuple(threshs)
    ws = torch.empty((int(lib.scf_cal_epe_workspace_bytes(n, h, w)),), dtype=torch.uint8, device=dev)
    E = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    P = lambda t: None if t is None else t.data_ptr()
    if reduction == 'none':
        err_map = E(n, h, w)
        _lib.check(lib.scf_cal_epe(pt, pp, pm, n, h, w, float(max_flow), None, 0, int(fix_threshold_quirk),
                                   P(err_map), None, None, None, None, ws.data_ptr(), ops._stream()), 'scf_cal_epe')
        return err_map
    acc = {}
    # the kernel counts up to 8 thresholds per pass: longer lists take one pass per group of 8 (the mean is the same
    # number every time)
    groups = [threshs[i:i + 8] for i in range(0, len(threshs), 8)] or [()]
    for grp in groups:
        thr = (C.c_float * max(len(grp), 1))(*[float(t) for t in grp])
        mean = ratios = tmean = tratios = None
        if reduction == 'mean':
            mean, ratios = E(n), E(max(len(grp), 1), n)
        else:
            tmean, tratios = E(1), E(max(len(grp), 1))
        _lib.check(lib.scf_cal_epe(pt, pp, pm, n, h, w, float(max_flow), thr, len(grp), int(fix_threshold_quirk),
                                   None, P(mean), P(ratios), P(tmean), P(tratios), ws.data_ptr(), ops._stream()),
                   'scf_cal_epe')
        if 'mean' not in acc:
            acc['mean'] = mean if reduction == 'mean' else tmean.reshape(())
        for i, t in enumerate(grp):
            acc[f'{t}px'] = ratios[i] if reduction == 'mean' else tratios[i]
    return acc


# ---------------------------------------------------------------- pose errors (8(f) row 4)
def eval_pose_error(verts_list, gt_t, gt_r, pred_t, pred_r, labels, k, symmetry_types,
                    mesh_diameters, device='cuda:0'):
    """``BaseDataset.eval_pose_error`` (datasets/base_dataset.py:378-424), same arguments
    (numpy arrays / lists) and returns: (error_3d_normalized, error_2d, error_3d) float64
    arrays.  ADD, ADD-S (``symmetry_types['cls_<label+1>']``) and the 2-D reprojection error
    run in one HIP launch per class (``scf_pose_error``, float64)."""
    import ctypes as C
    import numpy as np
    from . import _lib
    lib = _lib.load()
    n = len(gt_t)
    to = lambda a, shape: torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)
                                                               ).reshape(shape), device=device)
    gr, pr, kk = to(gt_r, (n, 9)), to(pred_r, (n, 9)), to(k, (n, 9))
    gt, pt = to(gt_t, (n, 3)), to(pred_t, (n, 3))
    e3 = torch.zeros(n, dtype=torch.float64, device=device)
    e2 = torch.zeros(n, dtype=torch.float64, device=device)
    labels = np.asarray(labels)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    keep = []
    for c in np.unique(labels):
        sel = np.nonzero(labels == c)[0].astype(np.int32)
        v = to(verts_list[c], (-1, 3))
        idx = torch.as_tensor(sel, device=device)
        keep += [v, idx]
        _lib.check(lib.scf_pose_error(v.data_ptr(), v.shape[0], gr.data_ptr(), gt.data_ptr(),
                                      pr.data_ptr(), pt.data_ptr(), kk.data_ptr(), idx.data_ptr(),
                                      len(sel), int(bool(symmetry_types.get(f'cls_{c + 1}', False))),
                                      e3.data_ptr(), e2.data_ptr(), stream), 'scf_pose_error')
    e3 = e3.cpu().numpy()
    diam = np.asarray([mesh_diameters[c] for c in labels], dtype=np.float64)
    return e3 / diam, e2.cpu().numpy(), e3


def eval_rot_error(gt_r, pred_r):
    """datasets/pose.py:106-112: geodesic angle in degrees (torch, any device)."""
    gt_r, pred_r = torch.as_tensor(gt_r), torch.as_tensor(pred_r)
    c = 0.5 * (torch.diagonal(pred_r @ torch.linalg.inv(gt_r), dim1=1, dim2=2).sum(-1) - 1.0)
    return torch.rad2deg(torch.arccos(c.clamp(-1.0, 1.0)))


def eval_tran_error(gt_t, pred_t):
    """datasets/pose.py:114-119 -> (|dt|, |dz|, |dxy|)."""
    gt_t, pred_t = torch.as_tensor(gt_t), torch.as_tensor(pred_t)
    return ((gt_t - pred_t).norm(dim=-1), (gt_t[:, -1] - pred_t[:, -1]).abs(),
            (gt_t[:, :2] - pred_t[:, :2]).norm(dim=-1))
