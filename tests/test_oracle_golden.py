"""CPU: the oracle (oracle/scflow_oracle.py) against golden vectors produced by
the reference's own source files (tests/golden/make_golden.py).

Tolerances (r6): <= 3 x the largest error measured over 1 / 2 / 4 / 8 / 16 host threads
(``pytest -s`` prints every ``[measured]`` line).  At the thread count the fixtures were written with the oracle
reproduces the reference BIT FOR BIT (every line reads 0.000e+00): what the tolerances cover is torch's own CPU
convolution summing in a thread-count-dependent order, not a difference between oracle and reference.  ``rtol`` is 0
unless stated: the oracle must not be looser than the kernels it judges (tests/test_gpu_*.py)."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from scflow_amd.synthetic import make_inputs
from scflow_amd.weights import fill_state_dict


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    return {k: (torch.from_numpy(d[k]) if d[k].dtype.kind in 'fi' and d[k].shape != () else d[k])
            for k in d.files}


def _close(a, b, atol, rtol=0.0, what=''):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    print(f'[measured] oracle vs fixture {what}: max abs err {float(err.max()):.3e} (atol {atol:g})')    # pytest -s
    assert bool((err <= lim).all()), f'{what}: max err {float(err.max()):.3e} (atol {atol})'


# whole-refiner outputs (px, px, -, mm, -, -, -); measured over 1..16 threads: 6.1e-5, 3.9e-5, 6.3e-8, 1.2e-4, 6.9e-7,
# 1.2e-7, 1.3e-7
_REFINER_TOL = dict(flow_from_pose=2e-4, flow_from_pred=1.2e-4, rotation=3e-7, translation=4e-4,
                    mask=2.5e-6, delta_rotation=4e-7, delta_translation=4e-7)


@pytest.mark.parametrize('name', ['corr_pyramid.npz', 'corr_pyramid_12x20.npz'])
def test_corr_pyramid(golden_dir, name):
    g = _load(golden_dir, name)
    pyr = oracle.correlation_pyramid(g['feat1'], g['feat2'], 4)
    for i, p in enumerate(pyr):
        _close(p, g[f'level{i}'], atol=2e-6, what=f'pyramid level {i}')          # measured 0


@pytest.mark.parametrize('name', ['corr_lookup.npz', 'corr_lookup_12x20.npz'])
def test_corr_lookup(golden_dir, name):
    g = _load(golden_dir, name)
    pyr = oracle.correlation_pyramid(g['feat1'], g['feat2'], 4)
    out = oracle.corr_lookup(pyr, g['flow'].clone(), 4)
    assert out.shape[1] == 4 * 81
    _close(out, g['out'], atol=2e-6, what='lookup')                              # measured 0


def test_corr_lookup_channel_order(golden_dir):
    """channel k = 81*l + 9*a + b samples x_off = a-4, y_off = b-4."""
    g = _load(golden_dir, 'corr_lookup_onehot.npz')
    pyr = oracle.correlation_pyramid(g['feat1'], g['feat2'], 4)
    out = oracle.corr_lookup(pyr, torch.zeros((1, 2, 16, 16)), 4)
    _close(out, g['out'], atol=1e-6, what='lookup one-hot')
    # query (x=8,y=8), target (x=5,y=10): x_off=-3 -> a=1, y_off=+2 -> b=6 -> k=15
    assert int(out[0, :81, 8, 8].argmax()) == 9 * 1 + 6
    assert abs(float(out[0, 15, 8, 8]) - 16.0 / 2.0) < 1e-4   # 4*4/sqrt(4)


def test_pose_math(golden_dir):
    g = _load(golden_dir, 'pose_math.npz')
    r, t = oracle.pose_from_delta_pose(g['d_rot'], g['d_trans'], g['rot'], g['trans'])
    _close(r, g['rot_new'], atol=2e-7, what='R new')                            # measured 0 (1 ulp = 6e-8)
    _close(t, g['trans_new'], atol=1e-4, what='t new (mm)')                     # measured 0 (1 ulp at 800 mm = 6e-5)
    pts = [oracle.unproject_depth(g['depth'][i], g['k'][i], g['rot'][i], g['trans'][i])
           for i in range(3)]
    assert [len(a) for a, _ in pts] == list(g['npts'])
    _close(pts[0][0], g['pts2d_0'], atol=0, what='pts2d')
    _close(pts[0][1], g['pts3d_0'], atol=1e-4, what='pts3d (mm)')               # measured 0
    for inv, key in ((0., 'flow_inv0'), (400., 'flow_inv400')):
        f = oracle.flow_from_pose_and_points(r, t, g['k'], [a for a, _ in pts],
                                             [b for _, b in pts], 32, 32, invalid_num=inv)
        _close(f, g[key], atol=2e-5, what=key)                                   # measured 0 (flows up to ~30 px)


@pytest.mark.parametrize('kind', ['IN', 'BN'])
def test_encoder(golden_dir, kind):
    g = _load(golden_dir, f'encoder_{kind}.npz')
    keys = json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))['shapes']
    pre = 'render_encoder.' if kind == 'IN' else 'context.'
    sd = fill_state_dict({k[len(pre):]: v for k, v in keys.items() if k.startswith(pre)}, seed=3)
    out = oracle.raft_encoder(g['x'], sd, '', kind)
    _close(out, g['out'], atol=6e-6, what=f'encoder {kind}')                    # measured <= 1.9e-6


def test_update_block(golden_dir):
    g = _load(golden_dir, 'update_block.npz')
    keys = json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))['shapes']
    shapes = {k[len('decoder.'):]: v for k, v in keys.items()
              if k.startswith(('decoder.encoder.', 'decoder.gru.', 'decoder.flow_pred.',
                               'decoder.mask_pred.'))}
    sd = fill_state_dict(shapes, seed=4)
    motion = oracle.motion_encoder(g['corr'], g['flow'], sd, 'encoder.')
    _close(motion, g['motion'], atol=4e-6, what='motion')                       # measured <= 1.3e-6
    h_new = oracle.sepconv_gru(g['h'], torch.cat([g['cxt'], motion], 1), sd, 'gru.')
    _close(h_new, g['h_new'], atol=2e-6, what='h_new')                          # measured <= 5.2e-7
    _close(oracle.xhead(h_new, sd, 'flow_pred.', 'flow'), g['d_flow'], atol=2.5e-6, what='d_flow')     # <= 7.2e-7
    _close(oracle.xhead(h_new, sd, 'mask_pred.', 'mask'), g['mask_logit'], atol=2e-6, what='mask logit')  # <= 4.2e-7


def test_pose_head_label_quirk(golden_dir):
    """mixed labels: every sample is decoded with class label[0]."""
    g = _load(golden_dir, 'pose_head.npz')
    keys = json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))['shapes']
    sd = fill_state_dict({k: v for k, v in keys.items() if k.startswith('decoder.pose_pred.')},
                         seed=4)
    x = torch.randn((3, 224, 32, 32), generator=torch.Generator().manual_seed(int(g['x_seed'])))
    r, t = oracle.multiclass_pose_head(x, g['label'], sd, 'decoder.pose_pred.')
    _close(r, g['rot'], atol=5e-7, what='pose head rot')                        # measured 0
    _close(t, g['trans'], atol=5e-7, what='pose head trans')
    r2, t2 = oracle.multiclass_pose_head(x, torch.tensor([2, 2, 2]), sd, 'decoder.pose_pred.')
    _close(r2, g['rot'], atol=5e-7, what='rot, all label[0]')       # == "all label[0]"
    r5, _ = oracle.multiclass_pose_head(x, torch.tensor([5, 5, 5]), sd, 'decoder.pose_pred.')
    _close(r5, g['rot_label5'], atol=5e-7, what='rot label 5')
    assert float((r5 - r).abs().max()) > 1e-4


def test_full_refiner(golden_dir):
    g = _load(golden_dir, 'refiner_full.npz')
    keys = json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))['shapes']
    sd = fill_state_dict(keys, seed=int(g['weight_seed']))
    inp = make_inputs(int(g['n']), 256, 256, seed=int(g['input_seed']))
    assert torch.equal(inp['label'], g['label'])
    with torch.no_grad():
        fr, fl, hf, cf = oracle.extract_feat(inp['render_images'], inp['real_images'], sd)
        _close(fr[:, ::8], g['feat_render'], atol=8e-6, what='feat_render')             # measured <= 2.4e-6
        _close(fl[:, ::8], g['feat_real'], atol=8e-6, what='feat_real')             # measured <= 2.4e-6
        _close(hf[:, ::8], g['h_feat'], atol=8e-6, what='h_feat')             # measured <= 2.4e-6
        _close(cf[:, ::8], g['cxt_feat'], atol=8e-6, what='cxt_feat')             # measured <= 2.4e-6
        outs = oracle.get_pose(inp['render_images'], inp['real_images'], inp['ref_rotation'],
                               inp['ref_translation'], inp['depth'], inp['internel_k'],
                               inp['label'], sd, iters=int(g['iters']))
    names = ['flow_from_pose', 'flow_from_pred', 'rotation', 'translation', 'mask',
             'delta_rotation', 'delta_translation']
    tol = _REFINER_TOL
    for nm, seq in zip(names, outs):
        st = torch.stack(list(seq))
        if st.dim() == 5:
            st = st[..., ::4, ::4]
        _close(st, g[nm], atol=tol[nm], what=nm)


def test_full_refiner_masked_branches(golden_dir):
    """decoder switches mask_flow / mask_corr (scflow_decoder.py:199-205; both False in the SCFlow
    config): the oracle against the reference run with both on."""
    g = _load(golden_dir, 'refiner_masked.npz')
    keys = json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))['shapes']
    sd = fill_state_dict(keys, seed=int(g['weight_seed']))
    inp = make_inputs(int(g['n']), 256, 256, seed=int(g['input_seed']))
    with torch.no_grad():
        outs = oracle.get_pose(inp['render_images'], inp['real_images'], inp['ref_rotation'],
                               inp['ref_translation'], inp['depth'], inp['internel_k'],
                               inp['label'], sd, iters=int(g['iters']), mask_flow=True, mask_corr=True)
    names = ['flow_from_pose', 'flow_from_pred', 'rotation', 'translation', 'mask',
             'delta_rotation', 'delta_translation']
    tol = _REFINER_TOL
    for nm, seq in zip(names, outs):
        st = torch.stack(list(seq))
        if st.dim() == 5:
            st = st[..., ::4, ::4]
        _close(st, g[nm], atol=tol[nm], what=nm)


def test_pose_math_linear_depth_transform(golden_dir):
    """pose.py:139-141: depth_transform other than 'exp' -> t_z * (d_z + 1)."""
    g = _load(golden_dir, 'pose_math_linear.npz')
    r, t = oracle.pose_from_delta_pose(g['d_rot'], g['d_trans'], g['rot'], g['trans'], depth_transform='linear')
    _close(r, g['rot_new'], atol=2e-7, what='R new (linear)')
    _close(t, g['trans_new'], atol=1e-4, what='t new (linear, mm)')
    _, t_exp = oracle.pose_from_delta_pose(g['d_rot'], g['d_trans'], g['rot'], g['trans'])
    assert float((t_exp - t).abs().max()) > 1e-2          # the two branches really differ on this input


def _refiner_fixture(golden_dir, name, keys, H, W, **kw):
    g = _load(golden_dir, name)
    sd = fill_state_dict(keys, seed=int(g['weight_seed']), shared_encoder=kw.pop('shared', True))
    inp = make_inputs(int(g['n']), H, W, seed=int(g['input_seed']))
    assert torch.equal(inp['label'], g['label'])
    with torch.no_grad():
        outs = oracle.get_pose(inp['render_images'], inp['real_images'], inp['ref_rotation'],
                               inp['ref_translation'], inp['depth'], inp['internel_k'],
                               inp['label'], sd, iters=int(g['iters']), **kw)
    names = ['flow_from_pose', 'flow_from_pred', 'rotation', 'translation', 'mask',
             'delta_rotation', 'delta_translation']
    for nm, seq in zip(names, outs):
        st = torch.stack(list(seq))
        if st.dim() == 5:
            st = st[..., ::4, ::4]
        _close(st, g[nm], atol=_REFINER_TOL[nm], what=f'{name} {nm}')


def test_refiner_separate_encoders_linear_depth(golden_dir):
    """seperate_encoder=True (own weights for the real-image encoder, base_refiner.py:33-35) and the decoder's
    depth_transform='linear': the oracle against the reference built with exactly those options."""
    keys = json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))['shapes']
    _refiner_fixture(golden_dir, 'refiner_options.npz', keys, 256, 256, shared=False, depth_transform='linear')


def test_refiner_512x640_feat_size(golden_dir):
    """SCFlow at 512 x 640 with the pose head's feat_size=(64, 80) (pose_head.py:121,147,162)."""
    keys = dict(json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))['shapes'])
    keys.update(json.load(open(os.path.join(golden_dir, 'state_dict_keys_512x640.json')))['shapes'])
    assert keys['decoder.pose_pred.fc_layers.0.0.weight'] == [1024, 10240]
    _refiner_fixture(golden_dir, 'refiner_512x640.npz', keys, 512, 640)


def test_refiner_conv_gru_radius3(golden_dir):
    """decoder gru_type='Conv' (one pass of 3x3 gates, raft_decoder.py:178-181) + radius=3 (196 correlation channels)."""
    keys = json.load(open(os.path.join(golden_dir, 'state_dict_keys_conv_gru_r3.json')))['shapes']
    assert keys['decoder.gru.conv_z.0.conv.weight'] == [128, 384, 3, 3] and 'decoder.gru.conv_z.1.conv.weight' not in keys
    assert keys['decoder.encoder.corr_net.0.conv.weight'][1] == 4 * 49
    _refiner_fixture(golden_dir, 'refiner_conv_gru_r3.npz', keys, 256, 256, radius=3)


def test_label_mode_per_sample_is_index_select_done_right(golden_dir):
    """oracle.multiclass_pose_head(label_mode=1) -- NOT the reference -- decodes sample n with class label[n]:
    equal, row by row, to the reference path called with that sample's label for the whole batch."""
    g = _load(golden_dir, 'pose_head.npz')
    keys = json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))['shapes']
    sd = fill_state_dict({k: v for k, v in keys.items() if k.startswith('decoder.pose_pred.')}, seed=4)
    x = torch.randn((3, 224, 32, 32), generator=torch.Generator().manual_seed(int(g['x_seed'])))
    label = g['label']
    assert len(set(label.tolist())) > 1
    r1, t1 = oracle.multiclass_pose_head(x, label, sd, 'decoder.pose_pred.', label_mode=1)
    for n in range(3):
        rn, tn = oracle.multiclass_pose_head(x, label[n].repeat(3), sd, 'decoder.pose_pred.')
        assert torch.equal(r1[n], rn[n]) and torch.equal(t1[n], tn[n])
    r0, _ = oracle.multiclass_pose_head(x, label, sd, 'decoder.pose_pred.')
    assert torch.equal(r0[0], r1[0]) and not torch.equal(r0[1], r1[1])
