"""Generate the golden fixtures in this directory from the REFERENCE's own
source files (``/root/reference``), executed unmodified on CPU.

Run once in the build container (the reference does not exist on the GPU
box; only the ``.npz`` / ``.json`` data written here travels):

    python tests/golden/make_golden.py

What pins what
--------------
* ``corr_pyramid.npz``, ``corr_lookup*.npz``, ``pose_math.npz`` come from
  ``CorrelationPyramid``, ``CorrLookup`` and ``models/utils/pose.py`` -- pure
  torch; only import-only stubs are involved (``pinned_under='reference
  source, import stubs only'``).
* ``encoder.npz``, ``update_block.npz``, ``pose_head.npz``,
  ``refiner_full.npz`` and ``state_dict_keys.json`` additionally depend on the
  hand-written mini-mmcv in ``_refshim.py`` (ConvModule / norm factories),
  which restates mmcv 1.3.16 and is unverified against the real package
  (``pinned_under='reference source under mini-mmcv shim'``).

Weights: ``scflow_amd.weights.fill_state_dict`` (pure function of key name and
shape) loaded into the reference model with ``load_state_dict(strict=True)``.
Inputs: ``scflow_amd.synthetic.make_inputs`` or seeded ``torch.randn`` stored
in the fixture itself when small.
"""
import json
import os
import runpy
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _refshim  # noqa: E402

_refshim.install()

from scflow_amd.synthetic import make_inputs  # noqa: E402
from scflow_amd.weights import fill_state_dict  # noqa: E402

from models.decoder.raft_decoder import (ConvGRU, CorrelationPyramid,  # noqa: E402
                                         MotionEncoder, XHead)
from models.encoder.raft_encoder import RAFTEncoder  # noqa: E402
from models.head.pose_head import MultiClassPoseHead  # noqa: E402
from models.refiner.builder import REFINERS  # noqa: E402
from models.refiner.scflow_refiner import SCFlowRefiner  # noqa: F401,E402
from models.utils.corr_lookup import CorrLookup  # noqa: E402
from models.utils.pose import (cal_3d_2d_corr, get_flow_from_delta_pose_and_points,  # noqa: E402
                               get_pose_from_delta_pose)
from mmcv.utils import build_from_cfg  # noqa: E402

STUBS = 'reference source, import stubs only'
SHIM = 'reference source under mini-mmcv shim (mmcv 1.3.16 restated, unverified)'


def save(name, pinned_under, **arrays):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
           for k, v in arrays.items()}
    out['pinned_under'] = np.asarray(pinned_under)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@torch.no_grad()
def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # -------------------------------------------------- correlation pyramid
    f1, f2 = rnd((2, 256, 8, 8), 11), rnd((2, 256, 8, 8), 12)
    pyr = CorrelationPyramid(num_levels=4)(f1, f2)
    save('corr_pyramid.npz', STUBS, feat1=f1, feat2=f2,
         **{f'level{i}': p for i, p in enumerate(pyr)})
    # odd-sized, C not a power of four (sqrt(C) inexact)
    g1, g2 = rnd((1, 96, 12, 20), 13), rnd((1, 96, 12, 20), 14)
    pyr = CorrelationPyramid(num_levels=4)(g1, g2)
    save('corr_pyramid_12x20.npz', STUBS, feat1=g1, feat2=g2,
         **{f'level{i}': p for i, p in enumerate(pyr)})

    # ------------------------------------------------------------ lookup
    for tag, (n, c, h, w) in {'': (2, 64, 16, 16), '_12x20': (1, 32, 12, 20)}.items():
        a, b = rnd((n, c, h, w), 21), rnd((n, c, h, w), 22)
        pyr = CorrelationPyramid(num_levels=4)(a, b)
        flow = rnd((n, 2, h, w), 23, scale=3.0)
        flow[0, :, 0, 0] = torch.tensor([-30., 2.])        # far out of bounds
        flow[0, :, 0, 1] = torch.tensor([0., 0.])          # exact integer taps
        flow[0, :, 0, 2] = torch.tensor([1.5, -2.25])
        flow[0, :, 1, 0] = torch.tensor([float(w), float(h)])
        out = CorrLookup(radius=4, align_corners=True)(pyr, flow.clone())
        save(f'corr_lookup{tag}.npz', STUBS, feat1=a, feat2=b, flow=flow, out=out)
    # channel-order probe: one-hot target -> which output channel lights up
    a = torch.zeros((1, 4, 16, 16)); b = torch.zeros((1, 4, 16, 16))
    a[0, 0, 8, 8] = 4.0; b[0, 0, 10, 5] = 4.0       # query (x=8,y=8) matches target (x=5,y=10)
    pyr = CorrelationPyramid(num_levels=4)(a, b)
    out = CorrLookup(radius=4, align_corners=True)(pyr, torch.zeros((1, 2, 16, 16)))
    save('corr_lookup_onehot.npz', STUBS, feat1=a, feat2=b, out=out)

    # --------------------------------------------------------- pose math
    inp = make_inputs(3, 32, 32, seed=5)
    inp['internel_k'][:, 0, 0] = 75.; inp['internel_k'][:, 1, 1] = 75.
    d_rot = torch.tensor([1., 0., 0., 0., 1., 0.]).repeat(3, 1) + rnd((3, 6), 31, 0.05)
    d_tr = rnd((3, 3), 32, 0.05)
    r_new, t_new = get_pose_from_delta_pose(d_rot, d_tr, inp['ref_rotation'],
                                            inp['ref_translation'], depth_transform='exp',
                                            detach_depth_for_xy=True)
    p2, p3 = [], []
    for i in range(3):
        a2, a3 = cal_3d_2d_corr(inp['depth'][i], inp['internel_k'][i], inp['ref_rotation'][i],
                                inp['ref_translation'][i])
        p2.append(a2); p3.append(a3)
    flow0 = get_flow_from_delta_pose_and_points(r_new, t_new, inp['internel_k'], p2, p3, 32, 32,
                                                invalid_num=0.)
    flow400 = get_flow_from_delta_pose_and_points(r_new, t_new, inp['internel_k'], p2, p3, 32,
                                                  32, invalid_num=400.)
    save('pose_math.npz', STUBS, depth=inp['depth'], k=inp['internel_k'],
         rot=inp['ref_rotation'], trans=inp['ref_translation'], d_rot=d_rot, d_trans=d_tr,
         rot_new=r_new, trans_new=t_new, flow_inv0=flow0, flow_inv400=flow400,
         npts=np.array([len(x) for x in p2]),
         pts2d_0=p2[0], pts3d_0=p3[0])

    # ------------------------------------------------------------ encoder
    x = torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(41))
    for kind in ('IN', 'BN'):
        enc = RAFTEncoder(in_channels=3, out_channels=256, net_type='Basic',
                          norm_cfg=dict(type=kind)).eval()
        sd = fill_state_dict({k: v.shape for k, v in enc.state_dict().items()}, seed=3)
        enc.load_state_dict(sd, strict=True)
        save(f'encoder_{kind}.npz', SHIM, x=x, out=enc(x))

    # ------------------------------------------------------- update block
    act = dict(type='ReLU')
    me = MotionEncoder(num_levels=4, radius=4, net_type='Basic', conv_cfg=None, norm_cfg=None,
                       act_cfg=act)
    gru = ConvGRU(128, 256, net_type='SeqConv')
    fh = XHead(128, [256], 2, x='flow')
    mh = XHead(128, [256], 1, x='mask')
    mods = {'encoder.': me, 'gru.': gru, 'flow_pred.': fh, 'mask_pred.': mh}
    shapes = {p + k: v.shape for p, m in mods.items() for k, v in m.state_dict().items()}
    sd = fill_state_dict(shapes, seed=4)
    for p, m in mods.items():
        m.load_state_dict({k[len(p):]: v for k, v in sd.items() if k.startswith(p)}, strict=True)
    corr = rnd((2, 324, 8, 8), 51); flow = rnd((2, 2, 8, 8), 52, 2.0)
    h = torch.tanh(rnd((2, 128, 8, 8), 53)); cxt = torch.relu(rnd((2, 128, 8, 8), 54))
    motion = me(corr, flow)
    h_new = gru(h, torch.cat([cxt, motion], dim=1))
    save('update_block.npz', SHIM, corr=corr, flow=flow, h=h, cxt=cxt, motion=motion,
         h_new=h_new, d_flow=fh(h_new), mask_logit=mh(h_new))

    # ---------------------------------------------------------- pose head
    ph = MultiClassPoseHead(num_class=21, in_channels=224, net_type='Basic',
                            rotation_mode='ortho6d',
                            norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                            act_cfg=dict(type='ReLU'))
    sd = fill_state_dict({'decoder.pose_pred.' + k: v.shape for k, v in ph.state_dict().items()},
                         seed=4)
    ph.load_state_dict({k[len('decoder.pose_pred.'):]: v for k, v in sd.items()}, strict=True)
    xin = rnd((3, 224, 32, 32), 61)
    label = torch.tensor([2, 5, 7])
    r, t = ph(xin, label)
    r_same, t_same = ph(xin, torch.tensor([5, 5, 5]))
    save('pose_head.npz', SHIM, label=label, rot=r, trans=t, rot_label5=r_same,
         trans_label5=t_same, x_seed=61)

    # ------------------------------------------------------- full refiner
    cfg = dict(runpy.run_path(_refshim.REFERENCE_ROOT + '/configs/refine_models/scflow.py')['model'])
    cfg['renderer'] = None                       # pytorch3d renderer: out of scope
    cfg['pose_loss_cfg'] = cfg['flow_loss_cfg']  # point-matching loss needs trimesh meshes
    model = build_from_cfg(cfg, REFINERS).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, 'state_dict_keys.json'), 'w') as f:
        json.dump({'pinned_under': SHIM, 'num_tensors': len(shapes),
                   'num_params_unique': int(sum(np.prod(s) for k, s in shapes.items()
                                                if not k.startswith('real_encoder.'))),
                   'shapes': {k: list(s) for k, s in shapes.items()}}, f, indent=0)
    sd = fill_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    iters = 3
    inp = make_inputs(3, 256, 256, seed=7)
    model.decoder.iters = iters
    outs = model.get_pose(inp['render_images'], inp['real_images'], inp['ref_rotation'],
                          inp['ref_translation'], inp['depth'], inp['internel_k'], inp['label'])
    names = ['flow_from_pose', 'flow_from_pred', 'rotation', 'translation', 'mask',
             'delta_rotation', 'delta_translation']
    arrays = {}
    for nm, seq in zip(names, outs):
        st = torch.stack(list(seq))              # (iters, N, ...)
        if st.dim() == 5:                        # dense maps: keep every 4th pixel
            st = st[..., ::4, ::4]
        arrays[nm] = st
    fr, fl, hf, cf = model.extract_feat(inp['render_images'], inp['real_images'])
    arrays.update(feat_render=fr[:, ::8], feat_real=fl[:, ::8], h_feat=hf[:, ::8],
                  cxt_feat=cf[:, ::8])
    save('refiner_full.npz', SHIM, iters=iters, input_seed=7, weight_seed=0, n=3,
         label=inp['label'], **arrays)


@torch.no_grad()
def main_masked():
    """the decoder's non-default branches mask_flow / mask_corr (scflow_decoder.py:199-205): the
    reference refiner with both switched on, same weights / inputs as refiner_full.npz."""
    cfg = dict(runpy.run_path(_refshim.REFERENCE_ROOT + '/configs/refine_models/scflow.py')['model'])
    cfg['renderer'] = None
    cfg['pose_loss_cfg'] = cfg['flow_loss_cfg']
    cfg['decoder'] = dict(cfg['decoder'], mask_flow=True, mask_corr=True)
    model = build_from_cfg(cfg, REFINERS).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(fill_state_dict(shapes, seed=0), strict=True)
    iters = 3
    inp = make_inputs(3, 256, 256, seed=7)
    model.decoder.iters = iters
    outs = model.get_pose(inp['render_images'], inp['real_images'], inp['ref_rotation'],
                          inp['ref_translation'], inp['depth'], inp['internel_k'], inp['label'])
    names = ['flow_from_pose', 'flow_from_pred', 'rotation', 'translation', 'mask',
             'delta_rotation', 'delta_translation']
    arrays = {}
    for nm, seq in zip(names, outs):
        st = torch.stack(list(seq))
        if st.dim() == 5:
            st = st[..., ::4, ::4]
        arrays[nm] = st
    save('refiner_masked.npz', SHIM, iters=iters, input_seed=7, weight_seed=0, n=3,
         label=inp['label'], mask_flow=1, mask_corr=1, **arrays)


@torch.no_grad()
def main_next():
    """fixtures of the SURVEY 8(f) rows: pose-free RAFT decoders, cal_epe."""
    from models.decoder.raft_decoder import RAFTDecoder
    from models.decoder.raft_decoder_mask import RAFTDecoderMask
    from models.utils.flow import cal_epe
    torch.manual_seed(0)
    kw = dict(net_type='Basic', num_levels=4, radius=4, iters=2,
              corr_lookup_cfg=dict(align_corners=True), gru_type='SeqConv',
              act_cfg=dict(type='ReLU'))
    f1, f2 = rnd((2, 256, 8, 8), 71), rnd((2, 256, 8, 8), 72)
    h = torch.tanh(rnd((2, 128, 8, 8), 73)); cxt = torch.relu(rnd((2, 128, 8, 8), 74))
    flow0 = rnd((2, 2, 8, 8), 75, 0.5)
    shapes = {}
    for name, cls in (('raft_decoder', RAFTDecoder), ('raft_decoder_mask', RAFTDecoderMask)):
        dec = cls(**dict(kw, corr_lookup_cfg=dict(align_corners=True))).eval()
        sh = {'decoder.' + k: tuple(v.shape) for k, v in dec.state_dict().items()}
        shapes[name] = {k: list(v) for k, v in sh.items()}
        sd = fill_state_dict(sh, seed=6)
        dec.load_state_dict({k[len('decoder.'):]: v for k, v in sd.items()}, strict=True)
        out = dec(f1, f2, flow0.clone(), h, cxt)
        if name == 'raft_decoder':
            save(name + '.npz', SHIM, feat1=f1, feat2=f2, flow0=flow0, h=h, cxt=cxt,
                 flows=torch.stack(out))
        else:
            save(name + '.npz', SHIM, feat1=f1, feat2=f2, flow0=flow0, h=h, cxt=cxt,
                 flows=torch.stack(out[0]), occs=torch.stack(out[1]))
    with open(os.path.join(HERE, 'raft_decoder_keys.json'), 'w') as f:
        json.dump({'pinned_under': SHIM, 'shapes': shapes}, f, indent=0)
    # cal_epe (pure torch)
    tgt, pred = rnd((3, 2, 16, 16), 81, 3.0), rnd((3, 2, 16, 16), 82, 3.0)
    tgt[0, :, :4] = 500.
    mask = (rnd((3, 16, 16), 83) > -0.5).float()
    res = {}
    for red in ('mean', 'total_mean'):
        acc = cal_epe(tgt.clone(), pred.clone(), mask, reduction=red)
        for k, v in acc.items():
            res[f'{red}_{k}'] = v
    res['none'] = cal_epe(tgt.clone(), pred.clone(), mask, reduction='none')
    res['mean_nomask'] = cal_epe(tgt.clone(), pred.clone(), None, reduction='mean')['mean']
    save('cal_epe.npz', STUBS, tgt=tgt, pred=pred, mask=mask, **res)


def main_gtflow():
    """fixtures of SURVEY 8(f) row 3, second half: ground-truth flow generation."""
    from models.utils.pose import get_flow_from_delta_pose_and_depth
    from models.utils.flow import filter_flow_by_mask
    inp = make_inputs(3, 48, 64, seed=21)
    g = torch.Generator().manual_seed(5)
    ang = torch.randn((3, 3), generator=g) * 0.05
    rot_dst = torch.matrix_exp(torch.stack([torch.tensor(
        [[0., -a[2], a[1]], [a[2], 0., -a[0]], [-a[1], a[0], 0.]]) for a in ang]))
    rot_dst = torch.bmm(rot_dst, inp['ref_rotation'])
    trans_dst = inp['ref_translation'] + torch.randn((3, 3), generator=g) * torch.tensor([8., 8., 20.])
    flow = get_flow_from_delta_pose_and_depth(inp['ref_rotation'], inp['ref_translation'], rot_dst,
                                              trans_dst, inp['depth'], inp['internel_k'], invalid_num=400)
    mask = (inp['depth'] > 0).float()
    mask[:, :, 40:] = 0.                      # part of the target silhouette missing
    out = {}
    for ac in (False, True):
        out[f'filtered_ac{int(ac)}'] = filter_flow_by_mask(flow.clone(), mask, invalid_num=400,
                                                          align_corners=ac)
    save('gt_flow.npz', STUBS, depth=inp['depth'], k=inp['internel_k'], rot_src=inp['ref_rotation'],
         trans_src=inp['ref_translation'], rot_dst=rot_dst, trans_dst=trans_dst, flow=flow,
         mask=mask, **out)


def main_poseerr():
    """fixtures of SURVEY 8(f) row 4: ADD / ADD-S / 2-D pose errors (numpy, float64)."""
    from datasets.base_dataset import BaseDataset
    from datasets.pose import eval_rot_error, eval_tran_error
    rs = np.random.RandomState(12)
    verts = [rs.randn(150, 3) * 40., rs.randn(210, 3) * 25., rs.randn(64, 3) * 60.]
    n = 7
    labels = np.array([0, 2, 1, 1, 0, 2, 1])

    def rot(a):
        a = np.asarray(a, dtype=np.float64)
        th = np.linalg.norm(a)
        kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]]) / th
        return np.eye(3) + np.sin(th) * kx + (1 - np.cos(th)) * kx @ kx

    gt_r = np.stack([rot(rs.randn(3)) for _ in range(n)])
    pred_r = np.stack([rot(rs.randn(3) * 0.05) @ r for r in gt_r])
    gt_t = np.stack([np.array([rs.randn() * 50, rs.randn() * 50, 800 + rs.rand() * 300]) for _ in range(n)])
    pred_t = gt_t + rs.randn(n, 3) * np.array([3., 3., 12.])
    k = np.tile(np.array([[600., 0, 128], [0, 600., 128], [0, 0, 1]]), (n, 1, 1))
    sym = {'cls_2': True}                       # label 1 is symmetric -> ADD-S
    diam = [180., 110., 260.]
    e3n, e2, e3 = BaseDataset.eval_pose_error(None, verts, gt_t, gt_r, pred_t, pred_r, labels, k, sym, diam)
    te, tz, txy = eval_tran_error(gt_t, pred_t)
    np.savez(os.path.join(HERE, 'pose_error.npz'), pinned_under=np.array(STUBS),
             verts0=verts[0], verts1=verts[1], verts2=verts[2], labels=labels, gt_r=gt_r, pred_r=pred_r,
             gt_t=gt_t, pred_t=pred_t, k=k, diam=np.array(diam), e3n=e3n, e2=e2, e3=e3,
             rot_err=eval_rot_error(gt_r, pred_r), t_err=te, tz_err=tz, txy_err=txy)
    print('pose_error.npz')


@torch.no_grad()
def main_options():
    """round 6: constructor options of the path that the default fixtures do not exercise.
    * ``pose_math_linear.npz``: ``get_pose_from_delta_pose`` with a depth_transform other than 'exp'
      (pose.py:139-141) -- pure torch.
    * ``refiner_options.npz``: the reference refiner with ``seperate_encoder=True`` (two feature
      encoders with their OWN weights, base_refiner.py:33-35) and the decoder's
      ``depth_transform='linear'``; N = 2, 2 iterations, 256 x 256.
    * ``refiner_512x640.npz``: the reference refiner at 512 x 640 with the pose head's
      ``feat_size=(64, 80)`` (pose_head.py:121,147,162 -- the only SCFlowDecoder route to a large
      map); N = 1, 2 iterations.
    * ``refiner_conv_gru_r3.npz``: decoder ``gru_type='Conv'`` and ``radius=3``; N = 2, 2 iterations."""
    inp = make_inputs(3, 32, 32, seed=5)
    d_rot = torch.tensor([1., 0., 0., 0., 1., 0.]).repeat(3, 1) + rnd((3, 6), 31, 0.05)
    d_tr = rnd((3, 3), 32, 0.05)
    r_new, t_new = get_pose_from_delta_pose(d_rot, d_tr, inp['ref_rotation'], inp['ref_translation'],
                                            depth_transform='linear', detach_depth_for_xy=True)
    save('pose_math_linear.npz', STUBS, rot=inp['ref_rotation'], trans=inp['ref_translation'],
         d_rot=d_rot, d_trans=d_tr, rot_new=r_new, trans_new=t_new)

    names = ['flow_from_pose', 'flow_from_pred', 'rotation', 'translation', 'mask',
             'delta_rotation', 'delta_translation']

    def run(cfg, n, H, W, iters, seed, shared):
        model = build_from_cfg(cfg, REFINERS).eval()
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(fill_state_dict(shapes, seed=0, shared_encoder=shared), strict=True)
        ii = make_inputs(n, H, W, seed=seed)
        model.decoder.iters = iters
        outs = model.get_pose(ii['render_images'], ii['real_images'], ii['ref_rotation'],
                              ii['ref_translation'], ii['depth'], ii['internel_k'], ii['label'])
        arrays = {}
        for nm, seq in zip(names, outs):
            st = torch.stack(list(seq))
            if st.dim() == 5:
                st = st[..., ::4, ::4]
            arrays[nm] = st
        return shapes, ii, arrays

    base = dict(runpy.run_path(_refshim.REFERENCE_ROOT + '/configs/refine_models/scflow.py')['model'])
    base['renderer'] = None
    base['pose_loss_cfg'] = base['flow_loss_cfg']

    cfg = dict(base, seperate_encoder=True)
    cfg['decoder'] = dict(cfg['decoder'], depth_transform='linear')
    shapes, ii, arrays = run(cfg, 2, 256, 256, 2, 31, False)
    assert not torch.equal(fill_state_dict(shapes, 0, shared_encoder=False)['real_encoder.conv1.weight'],
                           fill_state_dict(shapes, 0, shared_encoder=False)['render_encoder.conv1.weight'])
    save('refiner_options.npz', SHIM, iters=2, input_seed=31, weight_seed=0, n=2, label=ii['label'],
         seperate_encoder=1, depth_transform=np.array('linear'), **arrays)

    # the non-separable GRU (gru_type='Conv': one pass of 3x3 gates, raft_decoder.py:178-181) on a radius-3 lookup
    # (4 x 49 = 196 correlation channels into the motion encoder)
    cfg = dict(base)
    cfg['decoder'] = dict(cfg['decoder'], gru_type='Conv', radius=3)
    shapes, ii, arrays = run(cfg, 2, 256, 256, 2, 35, True)
    with open(os.path.join(HERE, 'state_dict_keys_conv_gru_r3.json'), 'w') as f:
        json.dump({'pinned_under': SHIM, 'shapes': {k: list(v) for k, v in shapes.items()}}, f, indent=0)
    save('refiner_conv_gru_r3.npz', SHIM, iters=2, input_seed=35, weight_seed=0, n=2, label=ii['label'],
         gru_type=np.array('Conv'), radius=3, **arrays)

    cfg = dict(base)
    cfg['decoder'] = dict(cfg['decoder'])
    cfg['decoder']['pose_head_cfg'] = dict(cfg['decoder']['pose_head_cfg'], feat_size=(64, 80))
    shapes, ii, arrays = run(cfg, 1, 512, 640, 2, 33, True)
    with open(os.path.join(HERE, 'state_dict_keys_512x640.json'), 'w') as f:
        json.dump({'pinned_under': SHIM,
                   'shapes': {k: list(v) for k, v in shapes.items() if 'pose_pred.fc_layers.0' in k}}, f, indent=0)
    save('refiner_512x640.npz', SHIM, iters=2, input_seed=33, weight_seed=0, n=1, label=ii['label'],
         feat_size=np.array([64, 80]), **arrays)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'poseerr':
        main_poseerr()
    elif len(sys.argv) > 1 and sys.argv[1] == 'masked':
        main_masked()
    elif len(sys.argv) > 1 and sys.argv[1] == 'next':
        main_next()
    elif len(sys.argv) > 1 and sys.argv[1] == 'gtflow':
        main_gtflow()
    elif len(sys.argv) > 1 and sys.argv[1] == 'options':
        main_options()
    else:
        main()
