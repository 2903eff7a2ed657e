"""GPU: size-independent properties of the hot path at BASELINE.json's FULL sizes (configs[2]:
batch 32, 256x256 -> h = w = 32, 8 iterations; configs[4] shape for the lookup), where the CPU
oracle would take minutes.  Each property is exact in real arithmetic; tolerances only cover
fp32 rounding.  All calls go through the C-ABI (scflow_amd.ops -> libscflow_hip.so)."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

import scflow_amd
from scflow_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rnd(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


@pytest.mark.parametrize('n,h,w', [(32, 32, 32), (2, 60, 80)])
def test_pyramid_is_symmetric_and_pooled(n, h, w):
    """corr(f1,f2)[n,i,j] = corr(f2,f1)[n,j,i]; level l+1 = 2x2 average pool of level l."""
    f1, f2 = rnd((n, 256, h, w), 1), rnd((n, 256, h, w), 2)
    a = ops.corr_build(f1, f2, 4)
    b = ops.corr_build(f2, f1, 1)
    hw = h * w
    ta = a[0].reshape(n, hw, hw)
    tb = b[0].reshape(n, hw, hw).transpose(1, 2)
    assert float((ta - tb).abs().max()) <= 2e-5 * float(ta.abs().max())
    for l in range(3):
        want = F.avg_pool2d(a[l], 2, 2)
        assert float((a[l + 1] - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))
    # the tiled levels hold the same numbers
    mask = ops.pyramid_layout(h, w, 4, 4)
    t = ops.corr_build(f1, f2, 4, tiled_levels=mask)
    for l in range(4):
        got = ops.untile_level(t[l], h >> l, w >> l) if (mask >> l) & 1 else t[l]
        assert torch.equal(got, a[l])


@pytest.mark.parametrize('n,h,w', [(32, 32, 32), (8, 60, 80)])
def test_lookup_is_linear_in_the_volume(n, h, w):
    """L(a P + b Q; flow) = a L(P; flow) + b L(Q; flow) for a fixed flow field."""
    q = n * h * w
    flow = rnd((n, 2, h, w), 3, 4.0)
    P = [rnd((q, 1, h >> l, w >> l), 10 + l) for l in range(4)]
    Q = [rnd((q, 1, h >> l, w >> l), 20 + l) for l in range(4)]
    mix = [2.0 * p - 0.5 * r for p, r in zip(P, Q)]
    lp, lq, lm = ops.corr_lookup(P, flow, 4), ops.corr_lookup(Q, flow, 4), ops.corr_lookup(mix, flow, 4)
    assert lm.shape == (n, 324, h, w)
    assert float((lm - (2.0 * lp - 0.5 * lq)).abs().max()) <= 2e-5
    # integer flow: the centre tap of level 0 is the map value at the displaced position
    fi = torch.zeros((n, 2, h, w), device=DEV)
    fi[:, 0] = 1.0
    centre = ops.corr_lookup(P, fi, 4)[:, 40]                    # k = 9*4 + 4: x_off = y_off = 0
    ys, xs = torch.meshgrid(torch.arange(h, device=DEV), torch.arange(w, device=DEV), indexing='ij')
    maps = P[0].reshape(n, h, w, h, w)
    tgt = torch.where(xs + 1 < w, maps[:, ys, xs, ys, (xs + 1).clamp(max=w - 1)], torch.zeros((), device=DEV))
    assert float((centre - tgt).abs().max()) <= 1e-5 * max(1.0, float(tgt.abs().max()))


def test_convolution_is_linear_and_shift_equivariant_at_full_size():
    """conv(a x + b y) = a conv(x) + b conv(y) - (a+b-1) bias;  shifting the input by whole
    pixels shifts the output (interior), on the encoder's 128x128 / batch-64 layer shape."""
    n, c, H, W = 64, 64, 128, 128
    wt, b = rnd((64, c, 3, 3), 5, 0.05), rnd((64,), 6, 0.1)
    pc = ops.PackedConv.from_weight(wt, b, padding=1)
    x, y = rnd((n, c, H, W), 7), rnd((n, c, H, W), 8)
    cx, cy = ops.conv2d(pc, x), ops.conv2d(pc, y)
    cm = ops.conv2d(pc, 1.5 * x - 0.25 * y)
    want = 1.5 * cx - 0.25 * cy - 0.25 * b.view(1, -1, 1, 1)
    assert float((cm - want).abs().max()) <= 3e-5 * max(1.0, float(want.abs().max()))
    xs = torch.roll(x, shifts=(3, 5), dims=(2, 3))
    cs = ops.conv2d(pc, xs)
    assert float((cs[:, :, 8:-8, 8:-8] - torch.roll(cx, (3, 5), (2, 3))[:, :, 8:-8, 8:-8]).abs().max()) <= 2e-5


def test_refiner_is_batch_permutation_equivariant_at_full_size(golden_dir):
    """configs[2]: 32 pairs, 8 iterations.  Pairs are independent end to end (same label for all,
    see the label[0] quirk), so permuting the batch permutes every output."""
    shapes = json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))['shapes']
    m = scflow_amd.build_refiner(scflow_amd.scflow_model_cfg(iters=8))
    m.load_state_dict(scflow_amd.fill_state_dict(shapes, seed=0), strict=True)
    m = m.to(DEV)
    inp = scflow_amd.make_inputs(32, 256, 256, seed=11)
    inp['label'][:] = 3
    d = {k: v.to(DEV) for k, v in inp.items()}
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(0)).to(DEV)

    def run(dd):
        o = m.get_pose(dd['render_images'], dd['real_images'], dd['ref_rotation'], dd['ref_translation'],
                       dd['depth'], dd['internel_k'], dd['label'])
        return o[0][-1], o[2][-1], o[3][-1]

    f0, r0, t0 = run(d)
    f1, r1, t1 = run({k: v[perm].contiguous() for k, v in d.items()})
    assert torch.isfinite(f0).all() and torch.isfinite(r0).all()
    # the tile decomposition does not depend on the sample index -> identical arithmetic
    assert torch.equal(f0[perm], f1) and torch.equal(r0[perm], r1) and torch.equal(t0[perm], t1)
    # rotations stay orthonormal after 8 compositions
    eye = torch.eye(3, device=DEV).expand(32, 3, 3)
    assert float((torch.bmm(r0, r0.transpose(1, 2)) - eye).abs().max()) <= 1e-4


def test_repeated_launches_are_bit_identical():
    """The LDS-DMA kernels synchronise by hand (vmcnt waits + barriers around asynchronous
    memory -> LDS copies): a missing wait shows up as run-to-run differences."""
    x, w, b = rnd((32, 128, 32, 32), 31), rnd((512, 128, 3, 3), 32, 0.03), rnd((512,), 33)
    pc = ops.PackedConv.from_weight(w, b, padding=1)
    ref = ops.conv2d(pc, x, act=ops.ACT_RELU).clone()
    for _ in range(50):
        assert torch.equal(ops.conv2d(pc, x, act=ops.ACT_RELU), ref)
    f1, f2 = rnd((32, 256, 32, 32), 34), rnd((32, 256, 32, 32), 35)
    flow = rnd((32, 2, 32, 32), 36, 3.0)
    pyr = ops.corr_build(f1, f2, 4, tiled_levels=1)
    want = ops.corr_lookup(pyr, flow, 4, tiled_levels=1).clone()
    for _ in range(50):
        assert torch.equal(ops.corr_lookup(pyr, flow, 4, tiled_levels=1), want)
        assert torch.equal(ops.corr_build(f1, f2, 1, tiled_levels=1)[0], pyr[0])


def test_launch_bound_timers_measure_the_kernel():
    """scf_timer_* / ops.lookup_timing / ops.time_first_kernel: a timer bound to a launch reports a
    positive duration no longer than a recorded-event pair around the same launch."""
    f1, f2 = rnd((8, 256, 32, 32), 41), rnd((8, 256, 32, 32), 42)
    flow = rnd((8, 2, 32, 32), 43, 2.0)
    pyr = ops.corr_build(f1, f2, 4, tiled_levels=1)
    out = ops.corr_lookup(pyr, flow, 4, tiled_levels=1)
    ops.lookup_timing(True)
    pairs = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        got = ops.corr_lookup(pyr, flow, 4, tiled_levels=1)
        b.record()
        pairs.append((a, b))
    us = ops.lookup_timing(False)
    assert torch.equal(got, out)                      # the timed entry point computes the same thing
    assert len(us) == 5 and all(1.0 < u < 1e4 for u in us)
    outer = [a.elapsed_time(b) * 1e3 for a, b in pairs]
    assert min(us) <= min(outer) + 1.0
    t = ops.time_first_kernel(lambda: ops.corr_build(f1, f2, 4, tiled_levels=1))
    assert 1.0 < t < 1e5


# ------------------------------------------------------------------------------------------------
# round 6: randomised sweeps (tools/lab/conv_fuzz.py, tools/lab/corr_fuzz.py; 1500 + 200 cases ran clean on the MI355X
# before these seeded subsets were pinned).  They exist for the dispatch corner cases hand-picked shapes miss: the one
# the first sweep found (a two-segment 1x1 layer whose segment boundary splits a channel chunk was refused) is pinned below.
# ------------------------------------------------------------------------------------------------
def _lab(name):
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'lab', name + '.py')
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_conv2d_random_sweep(seed):
    """80 random (shape, kernel size, stride, segments, bias / BN / residual / ReLU) convolutions per seed through scf_conv2d --
    every kernel family -- against a CPU fp64 convolution, 40 eps * sum|w||x|."""
    assert _lab('conv_fuzz').run(80, seed, verbose=False) == 0


@pytest.mark.parametrize('seed', [21, 22])
def test_corr_pair_random_sweep(seed):
    """40 random (map size, channels, radius 1..7, levels, layout) correlation builds + lookups per seed against the oracle."""
    assert _lab('corr_fuzz').run(40, seed, verbose=False) == 0


@pytest.mark.parametrize('n,cin,cout,c0,hw', [(8, 324, 4, 16, (30, 26)), (1, 72, 1, 8, (32, 32)), (2, 224, 126, 8, (16, 16))])
def test_conv2d_two_segments_off_chunk_boundary(n, cin, cout, c0, hw):
    """a channel concat whose boundary is not a multiple of the layer's channel chunk (1x1 layers stage 32 channels): no
    kernel takes it as two segments; ops.conv2d materialises the concatenation and runs the same layer on one (r6; the
    library itself answers SCF_EUNSUPPORTED)."""
    import torch.nn.functional as F
    from scflow_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn((n, cin, *hw), generator=g)
    w = torch.randn((cout, cin, 1, 1), generator=g) * (1.0 / cin) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    pc = ops.PackedConv.from_weight(w.to(DEV), b.to(DEV), stride=1, padding=(0, 0))
    xd = x.to(DEV)
    got = ops.conv2d(pc, xd[:, :c0], xd[:, c0:], act=ops.ACT_RELU)
    want = torch.relu(F.conv2d(x, w, b))
    assert float((got.cpu() - want).abs().max()) <= 2e-5
    assert torch.equal(got, ops.conv2d(pc, xd, act=ops.ACT_RELU))


@pytest.mark.parametrize('seed', [31, 32])
def test_convgru_random_sweep(seed):
    """40 random ConvGRU cells per seed (state 32..128, input 32..258 channels, both GRU types, maps 4..80 wide, with / without
    the hoisted context term) through scf_sepconv_gru[_ctx] against the oracle (tools/lab/gru_fuzz.py; 150 cases ran clean)."""
    assert _lab('gru_fuzz').run(40, seed, verbose=False) == 0
