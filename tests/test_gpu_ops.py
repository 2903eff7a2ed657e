"""GPU: every C-ABI operator against the CPU oracle / plain torch fp32 on the
same seeded inputs.  Tolerances are fp32 round-off (different summation order)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from scflow_amd import ops
from scflow_amd.synthetic import make_inputs

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, want, atol, rtol=1e-5, what=''):
    got = got.detach().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    err = (got - want).abs()
    lim = atol + rtol * want.abs()
    assert bool((err <= lim).all()), f'{what}: max err {float(err.max()):.3e} > atol {atol}'


# ------------------------------------------------------------ correlation
@pytest.mark.parametrize('n,c,h,w', [(2, 256, 8, 8), (1, 96, 12, 20), (1, 4, 16, 16),
                                     (2, 256, 32, 32), (1, 6, 8, 8)])
def test_corr_build(n, c, h, w):
    f1, f2 = rnd((n, c, h, w), 1), rnd((n, c, h, w), 2)
    want = oracle.correlation_pyramid(f1, f2, 4) if min(h, w) >= 8 else \
        oracle.correlation_pyramid(f1, f2, 3)
    got = ops.corr_build(f1.to(DEV), f2.to(DEV), len(want))
    for g, wv in zip(got, want):
        close(g, wv, atol=3e-5, what='corr level')


def test_corr_build_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'corr_pyramid.npz'))
    got = ops.corr_build(torch.from_numpy(g['feat1']).to(DEV), torch.from_numpy(g['feat2']).to(DEV), 4)
    for i, lv in enumerate(got):
        close(lv, torch.from_numpy(g[f'level{i}']), atol=3e-5, what=f'golden level{i}')


@pytest.mark.parametrize('n,h,w,r,L', [(2, 16, 16, 4, 4), (1, 12, 20, 4, 3), (3, 32, 32, 4, 4),
                                       (1, 16, 16, 3, 4), (1, 8, 24, 4, 2), (37, 4, 4, 2, 2), (5, 4, 8, 1, 2)])
def test_corr_lookup(n, h, w, r, L):
    f1, f2 = rnd((n, 32, h, w), 3), rnd((n, 32, h, w), 4)
    pyr = oracle.correlation_pyramid(f1, f2, L)
    flow = rnd((n, 2, h, w), 5, 3.0)
    flow[0, :, 0, 0] = torch.tensor([-40., 3.])
    flow[0, :, 0, 1] = 0.
    flow[0, :, 1, 1] = torch.tensor([float(w), float(h)])
    flow[0, :, 2, 2] = torch.tensor([1e9, -1e9])
    want = oracle.corr_lookup(pyr, flow.clone(), r)
    got = ops.corr_lookup([p.to(DEV) for p in pyr], flow.to(DEV), r)
    close(got, want, atol=5e-5, what='lookup')


def test_corr_lookup_golden_and_channel_order(golden_dir):
    for name in ('corr_lookup.npz', 'corr_lookup_12x20.npz', 'corr_lookup_onehot.npz'):
        g = np.load(os.path.join(golden_dir, name))
        f1, f2 = torch.from_numpy(g['feat1']), torch.from_numpy(g['feat2'])
        flow = torch.from_numpy(g['flow']) if 'flow' in g.files else torch.zeros((1, 2, 16, 16))
        pyr = ops.corr_build(f1.to(DEV), f2.to(DEV), 4)
        got = ops.corr_lookup(pyr, flow.to(DEV), 4)
        close(got, torch.from_numpy(g['out']), atol=5e-5, what=name)
    assert int(got[0, :81, 8, 8].argmax()) == 9 * 1 + 6      # x_off=-3 -> a=1, y_off=+2 -> b=6


def _edge_flow(n, h, w, seed, scale=4.0):
    flow = rnd((n, 2, h, w), seed, scale)
    flow[0, :, 0, 0] = torch.tensor([-40., 3.])
    flow[0, :, 1, 1] = torch.tensor([float(w), float(h)])
    flow[0, :, 2, 2] = torch.tensor([1e9, -1e9])
    flow[0, :, 3, 3] = torch.tensor([-3.5, -2.25])
    return flow


# mask None = the layout the decoders pick (ops.pyramid_layout); explicit masks also tile levels the
# rule would leave row-major (small maps, padded 30x40 / 15x20 / 4x4 levels)
@pytest.mark.parametrize('n,c,h,w,mask', [(2, 32, 16, 16, 0b0001), (1, 64, 12, 24, None), (2, 256, 32, 32, None),
                                          (1, 20, 20, 40, None), (1, 32, 60, 80, None), (1, 32, 60, 80, 0b1111),
                                          (2, 32, 32, 32, 0b1111), (1, 16, 24, 48, 0b0110), (1, 32, 44, 56, 0b0010)])
def test_corr_tiled_levels_are_a_permutation_and_look_up_identically(n, c, h, w, mask):
    """scf_corr_build_ex / scf_corr_lookup_ex with a tile mask: every tiled level is the reference
    map bit-for-bit after untiling (padding dropped), row-major levels are bit-identical, and the
    lookup output is bit-identical to the reference-layout lookup (same loads, same arithmetic)."""
    if mask is None:
        mask = ops.pyramid_layout(h, w, 4, 4)
        assert mask & 1, 'these shapes tile at least level 0'
    f1, f2 = rnd((n, c, h, w), 11).to(DEV), rnd((n, c, h, w), 12).to(DEV)
    ref = ops.corr_build(f1, f2, 4)
    til = ops.corr_build(f1, f2, 4, tiled_levels=mask)
    for l, (a, b) in enumerate(zip(til, ref)):
        if (mask >> l) & 1:
            assert tuple(a.shape[-2:]) == ops.level_storage_shape(h, w, l, True)
            assert torch.equal(ops.untile_level(a, h >> l, w >> l), b), f'level {l}'
        else:
            assert torch.equal(a, b), f'level {l}'
    flow = _edge_flow(n, h, w, 13).to(DEV)
    want = ops.corr_lookup(ref, flow, 4)
    got = ops.corr_lookup(til, flow, 4, tiled_levels=mask)
    assert torch.equal(got, want)
    close(got, oracle.corr_lookup([p.cpu() for p in ref], flow.cpu().clone(), 4), atol=5e-5,
          what='tiled lookup vs oracle')


def test_pyramid_layout_rule():
    """levels whose rows are >= 24 floats and that do not fit the 10x10 window are tiled."""
    assert ops.pyramid_layout(32, 32, 4, 4) == 0b0001          # 256x256 crops: level 1 is 16 wide
    assert ops.pyramid_layout(60, 80, 4, 4) == 0b0011          # 480x640: 60x80 and 30x40
    assert ops.pyramid_layout(16, 16, 4, 4) == 0
    assert ops.pyramid_layout(30, 40, 4, 2) == 0               # level 0 is not a whole number of tiles, level 1 is 20 wide
    assert ops.pyramid_layout(62, 96, 4, 3) == 0b0110          # 31x48 and 15x24 are tiled (padded to 32x48, 16x24)
    assert ops.level_storage_shape(60, 80, 1, True) == (32, 40)
    assert ops.level_storage_shape(60, 80, 2, True) == (16, 24)
    assert ops.level_storage_shape(60, 80, 2, False) == (15, 20)


def test_corr_tiled_rejects_unaligned_level0():
    f = rnd((1, 8, 10, 12), 1).to(DEV)
    with pytest.raises(Exception):
        ops.corr_build(f, f, 2, tiled_levels=1)
    with pytest.raises(Exception):          # a level list in the wrong storage shape is refused too
        ops.corr_lookup(ops.corr_build(f, f, 2), torch.zeros((1, 2, 10, 12), device=DEV), 4, tiled_levels=2)


@pytest.mark.parametrize('n,h,w,r,L,mask', [(1, 24, 24, 5, 3, 0), (2, 16, 16, 7, 2, 0), (1, 24, 32, 6, 3, 0b001),
                                            (1, 192, 192, 4, 4, 0), (1, 192, 192, 4, 2, 0b011), (1, 16, 16, 1, 5, 0)])
def test_corr_lookup_generic_kernel(n, h, w, r, L, mask):
    """operator-seam generality (CorrLookup(radius, ...) on any pyramid, corr_lookup.py:91-102):
    radius > 4 and maps of more than 32767 floats take the plain gather kernel -- same oracle, same
    tolerance, both layouts."""
    f1, f2 = rnd((n, 8, h, w), 3).to(DEV), rnd((n, 8, h, w), 4).to(DEV)
    ref = ops.corr_build(f1, f2, L)
    til = ops.corr_build(f1, f2, L, tiled_levels=mask) if mask else ref
    flow = _edge_flow(n, h, w, 5, 3.0)
    got = ops.corr_lookup(til, flow.to(DEV), r, tiled_levels=mask)
    assert got.shape == (n, L * (2 * r + 1) ** 2, h, w)
    want = oracle.corr_lookup([p.cpu() for p in ref], flow.clone(), r)
    # the reference normalises pixel coordinates to [-1, 1] and grid_sample maps them back
    # (corr_lookup.py:64-67): an fp32 round trip whose coordinate error grows with the map width
    # (~2e-5 px at 192); the kernels sample at the exact coordinate
    close(got, want, atol=5e-5 if max(h, w) <= 64 else 2e-4, what='generic lookup')


def test_corr_lookup_generic_matches_fast_kernel_bits():
    """the two kernels share the centre / weight / blend code: on a map the fast kernel takes at
    r = 4 and the generic one at r = 5, the r = 4 taps (the inner 9x9 of the 11x11 window at every
    level) are the same numbers."""
    n, h, w, L = 1, 32, 40, 3
    f1, f2 = rnd((n, 16, h, w), 21).to(DEV), rnd((n, 16, h, w), 22).to(DEV)
    pyr = ops.corr_build(f1, f2, L)
    flow = _edge_flow(n, h, w, 23).to(DEV)
    fast = ops.corr_lookup(pyr, flow, 4).reshape(n, L, 9, 9, h, w)
    gen = ops.corr_lookup(pyr, flow, 5).reshape(n, L, 11, 11, h, w)[:, :, 1:10, 1:10]
    assert torch.equal(fast, gen)


@pytest.mark.parametrize('r', [4, 5])
def test_corr_lookup_nan_and_inf_flow_like_grid_sample(r):
    """a query whose flow is NaN / inf gets NaN in ALL its taps (torch's CPU grid_sample, i.e. the
    reference path, returns NaN there); its neighbours are untouched; a huge finite flow reads
    zero padding."""
    n, h, w, L = 1, 16, 24, 3
    f1, f2 = rnd((n, 8, h, w), 31), rnd((n, 8, h, w), 32)
    pyr = oracle.correlation_pyramid(f1, f2, L)
    flow = rnd((n, 2, h, w), 33, 2.0)
    flow[0, 0, 2, 3] = float('nan')
    flow[0, 1, 4, 4] = float('nan')
    flow[0, :, 5, 5] = float('inf')
    flow[0, 1, 6, 6] = float('-inf')
    flow[0, 0, 7, 7] = 1e30
    want = oracle.corr_lookup(pyr, flow.clone(), r)
    got = ops.corr_lookup([p.to(DEV) for p in pyr], flow.to(DEV), r).cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    for (y, x) in ((2, 3), (4, 4), (5, 5), (6, 6)):
        assert bool(torch.isnan(got[0, :, y, x]).all())
    assert float(got[0, :, 7, 7].abs().max()) == 0.0
    ok = ~torch.isnan(want)
    assert float((got[ok] - want[ok]).abs().max()) <= 5e-5


def test_lookup_of_constant_volume_is_partition_of_unity():
    """size-independent property: a constant map sampled bilinearly gives the constant for
    fully in-range windows and 0 far outside."""
    n, h, w = 1, 32, 32
    pyr = [torch.full((n * h * w, 1, h >> l, w >> l), 2.5, device=DEV) for l in range(4)]
    flow = torch.zeros((n, 2, h, w), device=DEV)
    out = ops.corr_lookup(pyr, flow, 4).cpu()
    assert torch.allclose(out[0, :81, 16, 16], torch.full((81,), 2.5), atol=1e-6)
    flow[:] = 1000.
    assert float(ops.corr_lookup(pyr, flow, 4).abs().max()) == 0.0


# ------------------------------------------------------------------ conv
CONV_CASES = [
    # n, cin, cout, k, stride, pad, H, W
    (2, 3, 64, (7, 7), 2, (3, 3), 64, 64),
    (1, 64, 64, (3, 3), 1, (1, 1), 32, 32),
    (2, 64, 96, (3, 3), 2, (1, 1), 32, 32),
    (1, 64, 96, (1, 1), 2, (0, 0), 32, 32),
    (1, 96, 128, (3, 3), 2, (1, 1), 16, 16),
    (1, 128, 256, (1, 1), 1, (0, 0), 8, 8),
    (2, 324, 256, (1, 1), 1, (0, 0), 8, 8),
    (1, 256, 192, (3, 3), 1, (1, 1), 8, 8),
    (2, 2, 128, (7, 7), 1, (3, 3), 8, 8),
    (1, 256, 126, (3, 3), 1, (1, 1), 8, 8),
    (1, 384, 128, (1, 5), 1, (0, 2), 8, 8),
    (1, 384, 128, (5, 1), 1, (2, 0), 8, 8),
    (1, 256, 2, (3, 3), 1, (1, 1), 8, 8),
    (2, 1, 64, (3, 3), 1, (1, 1), 8, 8),
    (3, 224, 128, (3, 3), 2, (1, 1), 32, 32),
    (3, 128, 128, (3, 3), 2, (1, 1), 16, 16),
    (3, 128, 128, (3, 3), 2, (1, 1), 8, 8),
    (1, 16, 40, (3, 3), 1, (1, 1), 12, 20),
    (1, 64, 64, (3, 3), 1, (1, 1), 128, 128),
    # thin inputs (Cin <= 4): the tap-contracting kernel (conv_taps.hip), both tile widths, ragged tiles
    (8, 3, 64, (7, 7), 2, (3, 3), 128, 128),
    (32, 2, 128, (7, 7), 1, (3, 3), 32, 32),
    (32, 1, 64, (3, 3), 1, (1, 1), 32, 32),
    (3, 4, 40, (3, 3), 1, (1, 1), 12, 20),
    (2, 2, 96, (5, 5), 2, (2, 2), 30, 22),
    (1, 3, 64, (7, 7), 2, (3, 3), 480, 640),
    # more tiles than the chip holds blocks: the kernel walks several tiles per block (weights staged once, the next
    # tile's patch under the current tile's MFMAs); one channel block / three, even / ragged tile counts per block
    (64, 3, 64, (7, 7), 2, (3, 3), 256, 256),
    (24, 2, 96, (5, 5), 2, (2, 2), 126, 94),
    (37, 1, 64, (3, 3), 1, (1, 1), 44, 36),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d(case):
    n, cin, cout, k, s, p, H, W = case
    x = rnd((n, cin, H, W), 10)
    wt = rnd((cout, cin, *k), 11, (1.0 / (cin * k[0] * k[1])) ** 0.5)
    b = rnd((cout,), 12, 0.1)
    want = torch.relu(F.conv2d(x, wt, b, stride=s, padding=p))
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), stride=s, padding=p)
    got = ops.conv2d(pc, x.to(DEV), act=ops.ACT_RELU)
    close(got, want, atol=2e-5, what=str(case))


# shapes large enough for the LDS-DMA kernel (>= 256 blocks), one per tile shape / chunk regime
DMA_CASES = [
    # n, cin, cout, k, pad, H, W                      expected tile
    (32, 64, 64, (3, 3), (1, 1), 64, 64),           # (2,2)
    (16, 96, 96, (3, 3), (1, 1), 64, 64),           # (3,1), > 64 KB LDS
    (32, 128, 128, (1, 5), (0, 2), 32, 32),         # (2,1), G=2
    (32, 128, 128, (5, 1), (2, 0), 32, 32),         # G=2, tall window
    (32, 60, 126, (3, 3), (1, 1), 32, 32),          # channel tail (60 = 7*8 + 4), ragged couts
    (64, 24, 40, (3, 3), (1, 1), 24, 40),           # (1,1), ragged tiles (Wo = 40, Ho = 24)
    (32, 224, 128, (3, 3), (1, 1), 32, 32, 2),      # stride 2 (pose head c0): parity-split patch rows
    (64, 64, 96, (3, 3), (1, 1), 64, 64, 2),        # stride 2, FC = 32
    (64, 32, 64, (3, 3), (1, 1), 30, 22, 2),        # stride 2, odd sizes, ragged tiles (W % 4 != 0: dword staging)
    (32, 324, 256, (1, 1), (0, 0), 32, 32),         # dense 1x1 on a full grid: 32-channel chunks, short last chunk (4)
    (64, 40, 64, (3, 3), (1, 1), 30, 22),           # stride 1, W % 4 != 0: dword staging, ragged everything
    (32, 72, 96, (3, 3), (1, 1), 16, 48, 2),        # stride 2 with the aligned-x4 staging, FC = 16, wide image
    (48, 64, 64, (1, 5), (0, 2), 20, 36),           # 1x5 with x4 staging: px_off = 2, tiles cut by the right edge
    (8, 128, 128, (3, 3), (1, 1), 60, 80),          # Wo = 80: 16-column fragments (5 x 16 instead of 3 x 32)
    (8, 64, 96, (5, 1), (2, 0), 28, 40),            # Wo = 40: 8-column fragments, 4 rows per fragment
    (8, 96, 64, (3, 3), (1, 1), 120, 160, 2),       # stride 2 onto a 60 x 80 map
    (32, 64, 32, (3, 3), (1, 1), 32, 32),           # small grid (256 pixel-split blocks): K-split tile, 1024 blocks
    (48, 48, 32, (3, 3), (1, 1), 25, 32),           # the same with an odd number of rows
    (40, 40, 64, (1, 5), (0, 2), 18, 24),           # K-split with 16-channel chunks of a 1x5 layer, Wo = 24
    # 1x1 / stride-2 shortcuts (resnet.py:721-730): a dense 1x1 over every second row / column (dilated gather)
    (64, 64, 96, (1, 1), (0, 0), 128, 128, 2),      # full grid, (3,1) tile, 2 chunks of 32 channels
    (32, 96, 128, (1, 1), (0, 0), 64, 64, 2),       # (2,1) tile, 3 chunks
    (8, 72, 96, (1, 1), (0, 0), 33, 41, 2),         # odd sizes (Ho = 17, Wo = 21), short last chunk, ragged tiles
    (2, 64, 96, (1, 1), (0, 0), 128, 128, 2),       # small grid: K-split tile on the dilated gather
    (1, 96, 128, (1, 1), (0, 0), 60, 80, 2),        # batch 1, 30 x 40 map
]


@pytest.mark.parametrize('case', DMA_CASES)
def test_conv2d_dma_kernel(case):
    import ctypes as C
    n, cin, cout, k, p, H, W = case[:7]
    st = case[7] if len(case) > 7 else 1
    x = rnd((n, cin, H, W), 30)
    wt = rnd((cout, cin, *k), 31, (1.0 / (cin * k[0] * k[1])) ** 0.5)
    b = rnd((cout,), 32, 0.1)
    want = F.conv2d(x, wt, b, stride=st, padding=p)
    res = rnd(tuple(want.shape), 33)
    want = torch.relu(want + res)
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), stride=st, padding=p)
    assert pc.wp4 is not None
    got = ops.conv2d(pc, x.to(DEV), res=res.to(DEV), act=ops.ACT_RELU)
    close(got, want, atol=3e-5, what=str(case))
    # the register-staged kernel must give the same numbers up to summation order
    pc2 = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), stride=st, padding=p, dma_packing=False)
    assert pc2.wp4 is None
    ref = ops.conv2d(pc2, x.to(DEV), res=res.to(DEV), act=ops.ACT_RELU)
    close(got, ref.cpu(), atol=3e-5, what='dma vs register-staged ' + str(case))


@pytest.mark.parametrize('case', [(1, 256, 256, (1, 5), (0, 2), 32, 32, 1, 128), (1, 128, 256, (3, 3), (1, 1), 32, 32, 1, 0),
                                  (2, 256, 128, (5, 1), (2, 0), 32, 32, 1, 128), (1, 324, 256, (1, 1), (0, 0), 32, 32, 1, 0),
                                  (1, 224, 128, (3, 3), (1, 1), 32, 32, 2, 128), (1, 104, 64, (3, 3), (1, 1), 20, 24, 1, 0),
                                  (1, 128, 128, (3, 3), (1, 1), 8, 8, 2, 0)])
def test_conv2d_ksplit_wave_groups(case):
    """grids with at most one K-split block per CU (batch 1): two wave groups per block walk alternate channel chunks
    (conv_dma_kernel NG = 2) -- against torch and against the one-group kernel (scf_tune dma_ksplit_groups = 1): same
    products, another fixed summation order"""
    n, cin, cout, k, p, H, W, st, c0 = case
    x = rnd((n, cin, H, W), 230)
    wt = rnd((cout, cin, *k), 231, (1.0 / (cin * k[0] * k[1])) ** 0.5)
    b = rnd((cout,), 232, 0.1)
    want = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), stride=st, padding=p)).float()
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), stride=st, padding=p)
    xd = x.to(DEV)
    x0, x1 = (xd[:, :c0], xd[:, c0:]) if c0 else (xd, None)
    prev = ops.set_conv_winograd(False)
    try:
        got2 = ops.conv2d(pc, x0, x1, act=ops.ACT_RELU)
        ops.tune('dma_ksplit_groups', 1)
        got1 = ops.conv2d(pc, x0, x1, act=ops.ACT_RELU)
    finally:
        ops.tune('dma_ksplit_groups', 0)
        ops.set_conv_winograd(prev)
    close(got2, want, atol=1e-5, what=f'two wave groups {case}')
    close(got1, want, atol=1e-5, what=f'one wave group {case}')
    if k in ((1, 5), (1, 1)) and n * ((H // st + 0) * (W // st) // 32) * ((cout + 31) // 32) <= 256:
        # small stages (four fit the LDS): these run two wave groups; 3x3 / 5x1 stages are too big and keep one
        assert float((got1 - got2).abs().max()) > 0.0, 'identical bits: the two-group kernel did not run'
    prev2 = ops.set_conv_winograd(False)
    again = ops.conv2d(pc, x0, x1, act=ops.ACT_RELU)
    ops.set_conv_winograd(prev2)
    assert torch.equal(again, got2)                  # deterministic


WINO_CASES = [
    # n, cin, cout, H, W, (c0 split or 0), BN, residual, relu -- every case is >= 128 blocks (the dispatch keeps
    # smaller grids on the direct kernels)
    (16, 64, 64, 32, 32, 0, False, False, True),
    (2, 128, 512, 32, 32, 0, False, False, True),      # XHead hidden layer shape
    (8, 256, 126, 32, 32, 192, False, False, True),    # motion encoder out conv: two segments, 126 channels
    (3, 96, 96, 64, 64, 0, True, True, True),          # odd fragment count; BN + residual
    (1, 64, 64, 128, 128, 0, False, False, False),     # encoder layer, no activation (InstanceNorm follows)
    (24, 30, 40, 20, 28, 0, False, True, False),       # ragged: Cin % 4 != 0, Cout % 32 != 0, Wo % 32 != 0
    (4, 128, 64, 60, 80, 0, False, False, True),       # 60 x 80 map (480 x 640 crops): narrower tile groups
    (128, 16, 32, 7, 10, 0, False, False, True),       # odd height, tiny width, rows not 16-byte aligned (dword patch copies)
    (64, 16, 64, 16, 16, 0, False, False, True),       # 4 chunks: the shortest K the kernel takes
    (5, 64, 96, 64, 64, 0, False, True, True),          # 240 tiles on 512 block slots / odd fragment count
    (40, 32, 64, 48, 32, 0, False, False, True),        # 960 tiles: persistent blocks walk 2 tiles each (one of them partially)
]


@pytest.mark.parametrize('variant', [1, 2])
@pytest.mark.parametrize('case', WINO_CASES)
def test_conv2d_winograd(case, variant):
    """F(2x2, 3x3) kernels vs torch fp64 and vs the direct kernel: same fp32 arithmetic with re-associated
    sums, so the error budget is a small multiple of the direct kernel's.  variant 1 = the pair kernel (a wave pair
    per channel fragment) forced on every case, variant 2 = the dispatch's choice: the quarter-domain kernel
    (one transform row x two channel fragments per wave) where the fragment count is even, else the pair kernel."""
    import ctypes as C
    n, cin, cout, H, W, c0, bn, with_res, relu = case
    x = rnd((n, cin, H, W), 40)
    wt = rnd((cout, cin, 3, 3), 41, (1.0 / (cin * 9)) ** 0.5)
    b = rnd((cout,), 42, 0.1)
    want = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    bnp = None
    if bn:
        gamma, beta = rnd((cout,), 43) * 0.2 + 1, rnd((cout,), 44) * 0.1
        mean, var = rnd((cout,), 45) * 0.1, rnd((cout,), 46).abs() * 0.5 + 0.5
        want = F.batch_norm(want, mean.double(), var.double(), gamma.double(), beta.double(), False, 0., 1e-5)
        bnp = [t.to(DEV) for t in (gamma, beta, mean, var)]
    res = rnd(tuple(want.shape), 47) if with_res else None
    if with_res:
        want = want + res.double()
    if relu:
        want = torch.relu(want)
    want = want.float()
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=1, bn=bnp)
    assert pc.wwino is not None
    # the C packer and the torch packer agree bit for bit
    lib = ops._lib.load()
    size = lib.scf_pack_conv_weight_wino_size(cout, cin)
    assert size == pc.wwino.numel()
    host = torch.empty(size)
    wc = wt.contiguous()
    assert lib.scf_pack_conv_weight_wino(wc.data_ptr(), cout, cin, host.data_ptr()) == 0
    assert torch.equal(host, pc.wwino.cpu())
    xd = x.to(DEV)
    kw = dict(res=None if res is None else res.to(DEV), act=ops.ACT_RELU if relu else ops.ACT_NONE)
    x0, x1 = (xd[:, :c0], xd[:, c0:]) if c0 else (xd, None)
    prev = ops.set_conv_winograd(False)
    try:
        direct = ops.conv2d(pc, x0, x1, **kw)
        ops.set_conv_winograd(True)
        ops.tune('wino_variant', variant)
        d, _ = ops.conv_desc(pc, x0, x1, **kw)
        info = (C.c_int32 * 4)()
        assert lib.scf_conv2d_query(C.byref(d), info) == 0
        assert info[3] < 0 and info[0] == 16, list(info)     # the Winograd kernel is the one that runs
        with ops.record_conv_kernels() as ran:
            got = ops.conv2d(pc, x0, x1, **kw)
        # the quarter-domain kernel needs an even fragment count and its own grid of >= CUs / 2 blocks: the ragged
        # 20 x 28 case gives it 120 blocks (5 strips of 4 rows) where the pair kernel has 144 (3 strips of 8 rows x 2
        # fragments), so that one stays on the pair kernel
        even_frags = ((cout + 31) // 32) % 2 == 0
        quarter = variant == 2 and even_frags and (H, W) != (20, 28)
        assert [k_ for _, k_ in ran] == ['winograd-q' if quarter else 'winograd'], ran
    finally:
        ops.tune('wino_variant', 0)
        ops.set_conv_winograd(prev)
    e_dir = float((direct.cpu() - want).abs().max())
    e_win = float((got.cpu() - want).abs().max())
    print(f'winograd {case} variant {variant}: max err {e_win:.2e} (direct kernel {e_dir:.2e})')
    close(got, want, atol=1e-5, what='winograd ' + str(case))      # measured <= 3.5e-6
    # small grids stay on the direct kernels
    xs = xd[:1, :, :8, :10].contiguous() if W >= 10 and H >= 8 else xd[:1]
    d, _ = ops.conv_desc(pc, xs if not c0 else xs[:, :c0], None if not c0 else xs[:, c0:])
    assert lib.scf_conv2d_query(C.byref(d), info) == 0 and info[0] != 16, list(info)


WINO1D_CASES = [
    # n, cin, cout, k, H, W, c0 split -- every case is >= 128 blocks
    (8, 256, 256, (1, 5), 32, 32, 128),     # GRU z|r, horizontal pass
    (16, 256, 128, (5, 1), 32, 32, 128),    # GRU q, vertical pass
    (32, 40, 64, (1, 5), 20, 30, 0),        # ragged: Cin % 8 != 0, Wo % 4 != 0 (dword patch copies), odd tile rows
    (24, 48, 64, (5, 1), 21, 28, 0),        # vertical, odd height
    (2, 64, 128, (1, 5), 60, 80, 0),        # 60 x 80 maps
    (2, 64, 128, (5, 1), 60, 80, 0),
    (24, 44, 64, (5, 1), 24, 32, 0),        # 11 chunks of 4 (F(4, 5)), 5.5 of 8 (F(2, 5))
    (20, 172, 128, (1, 5), 24, 32, 128),    # two input segments, a short last chunk in the second
]


@pytest.mark.parametrize('form', ['F(2,5)', 'F(4,5)'])
@pytest.mark.parametrize('case', WINO1D_CASES)
def test_conv2d_winograd_1d(case, form):
    """F(2, 5) kernel (conv_wino1d.hip) and F(4, 5) kernel (conv_wino1d4.hip; on every grid here: the ragged
    cases are below its dispatch threshold) vs torch fp64 and vs the direct kernel, plain epilogue with residual
    + ReLU; the GRU epilogues on them are covered by test_sepconv_gru_winograd."""
    import ctypes as C
    n, cin, cout, k, H, W, c0 = case
    f4 = form == 'F(4,5)'
    pad = (0, 2) if k == (1, 5) else (2, 0)
    x = rnd((n, cin, H, W), 50)
    wt = rnd((cout, cin, *k), 51, (1.0 / (cin * 5)) ** 0.5)
    b = rnd((cout,), 52, 0.1)
    want = F.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    res = rnd(tuple(want.shape), 53)
    want = torch.relu(want + res.double()).float()
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=pad)
    assert pc.wwino1d is not None
    lib = ops._lib.load()
    host = torch.empty(lib.scf_pack_conv_weight_wino1d_size(cout, cin))
    wc = wt.reshape(cout, cin, 5).contiguous()
    assert lib.scf_pack_conv_weight_wino1d(wc.data_ptr(), cout, cin, host.data_ptr()) == 0
    assert torch.equal(host, pc.wwino1d.cpu())
    host4 = torch.empty(lib.scf_pack_conv_weight_wino1d4_size(cout, cin))
    assert lib.scf_pack_conv_weight_wino1d4(wc.data_ptr(), cout, cin, host4.data_ptr()) == 0
    assert torch.equal(host4, pc.wwino1d4.cpu())
    xd = x.to(DEV)
    x0, x1 = (xd[:, :c0], xd[:, c0:]) if c0 else (xd, None)
    kw = dict(res=res.to(DEV), act=ops.ACT_RELU)
    prev = ops.set_conv_winograd(False)
    try:
        direct = ops.conv2d(pc, x0, x1, **kw)
        ops.set_conv_winograd(True)
        ops.tune('wino1d4', 2 if f4 else 0)
        d, _ = ops.conv_desc(pc, x0, x1, **kw)
        info = (C.c_int32 * 4)()
        assert lib.scf_conv2d_query(C.byref(d), info) == 0
        assert info[3] < 0 and info[0] == (8 if f4 else 6) and info[2] >= (64 if f4 else 128), list(info)
        with ops.record_conv_kernels() as ran:
            got = ops.conv2d(pc, x0, x1, **kw)
        assert [kk for _, kk in ran] == ['winograd ' + form], ran
    finally:
        ops.tune('wino1d4', 1)
        ops.set_conv_winograd(prev)
    e_dir = float((direct.cpu() - want).abs().max())
    e_win = float((got.cpu() - want).abs().max())
    print(f'winograd {form} {case}: max err {e_win:.2e} (direct kernel {e_dir:.2e})')
    close(got, want, atol=2e-5, what=f'winograd {form} ' + str(case))


@pytest.mark.parametrize('seed', range(12))
def test_winograd_kernels_random_shapes(seed):
    """Seeded random layer shapes (channels, map sizes, two segments, residual / BN / ReLU on and off) through
    both Winograd kernels vs the direct kernels: edge tiles, short last chunks, odd sizes, unaligned rows."""
    import ctypes as C
    import random
    rng = random.Random(1000 + seed)
    lib = ops._lib.load()
    for kind in ('2d', '1dh', '1dv', '1dh4', '1dv4'):
        f4 = kind.endswith('4')          # F(4, 5) on every grid (2), else F(2, 5) only (0)
        ops.tune('wino1d4', 2 if f4 else 0)
        kind = kind[:3]
        cin = rng.choice([16, 20, 36, 64, 72, 130])
        cout = rng.choice([64, 128, 192]) if kind != '2d' else rng.choice([32, 40, 64, 96, 126, 160])
        H, W = rng.choice([(16, 16), (20, 28), (23, 30), (32, 32), (33, 40), (48, 36), (12, 64)])
        k, pad = {'2d': ((3, 3), 1), '1dh': ((1, 5), (0, 2)), '1dv': ((5, 1), (2, 0))}[kind]
        n = 2
        x = rnd((n, cin, H, W), 60 + seed)
        wt = rnd((cout, cin, *k), 61 + seed, (1.0 / (cin * k[0] * k[1])) ** 0.5)
        b = rnd((cout,), 62 + seed, 0.1)
        pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=pad)
        # enough samples for the dispatch's grid threshold
        ops.set_conv_winograd(True)
        info = (C.c_int32 * 4)()
        while True:
            xd = rnd((n, cin, H, W), 63 + seed).to(DEV)
            d, _ = ops.conv_desc(pc, xd)
            assert lib.scf_conv2d_query(C.byref(d), info) == 0
            if info[0] in (16, 6, 8) or n >= 256:
                break
            n *= 2
        if info[0] not in (16, 6, 8):
            continue                     # e.g. a vertical pass on rows that are not 16-byte aligned: direct kernels
        if f4 and info[0] == 6:
            assert cin <= 16, (kind, cin, list(info))     # fewer than the five chunks the F(4, 5) kernel peels: F(2, 5)
            continue
        assert (info[0] == 8) == f4, (kind, f4, list(info))
        c0 = rng.choice([0, 0, 8, 16]) if cin > 16 and (kind == '2d' or cin % 8 == 0) else 0
        res = rnd((n, cout, H, W), 64 + seed).to(DEV) if rng.random() < 0.5 else None
        act = ops.ACT_RELU if rng.random() < 0.5 else ops.ACT_NONE
        x0, x1 = (xd[:, :c0], xd[:, c0:]) if c0 else (xd, None)
        got = ops.conv2d(pc, x0, x1, res=res, act=act)
        prev = ops.set_conv_winograd(False)
        try:
            want = ops.conv2d(pc, x0, x1, res=res, act=act)
        finally:
            ops.set_conv_winograd(True)
        err = float((got - want).abs().max())
        assert err <= 3e-5, (kind, f4, n, cin, cout, H, W, c0, err)
        assert err > 0.0, 'the two paths gave identical bits: the Winograd kernel did not run'
    ops.tune('wino1d4', 1)


# Winograd error scales with |input| and |weight|, not with |output|: the transforms add and subtract the RAW
# operands (B^T d B scales by up to 4, F(2, 5)'s points 0, +-1, +-2, inf by up to 10; G by 1/2 ... 1/24) before
# the products cancel.  N(0,1) operands hide that.  Stress operands: a DC offset far above
# the signal (x = 50 + N(0,1)), post-ReLU |N(0,1)| activations, weights spread over three decades.  The budget is
# stated in units of eps * sum|w||x| (eps = 2^-24: the forward error scale of ANY fp32 evaluation of the sum),
# max over all outputs; budgets = ~2x the worst value measured on the MI355X (gpurun r4a, 24 cases):
#   direct kernels  measured 0.9 ... 11.7  (an fma chain over K = Cin * taps = 576 ... 2304 terms; worst: wide weights)
#   F(2x2, 3x3)     measured 1.9 ... 5.8   -> budget 12: NOT worse than the direct chain on these operands (the
#                   transforms shorten the chain 2.25x; their own amplification is a factor <= 4)
#   F(2, 5)         measured 4.6 ... 8.7, wide-range weights 21.7 ... 29.9 -> budget 60 = 3.6e-6 relative to
#                   sum|w||x| (G scales the taps by 1/24 ... 1/4 before they recombine): 2.5x the direct kernels' worst
# i.e. max abs errors of 3e-6 (post-ReLU operands) ... 2.8e-3 on outputs of magnitude 5 ... 1400.
WINO_STRESS_OPERANDS = ['dc50', 'relu', 'wide_weights', 'dc50_wide']


def _stress_operands(kind, shape_x, shape_w, seed):
    x = rnd(shape_x, seed)
    fan = shape_w[1] * shape_w[2] * shape_w[3]
    wt = rnd(shape_w, seed + 1, (1.0 / fan) ** 0.5)
    if kind in ('dc50', 'dc50_wide'):
        x = x + 50.0
    if kind == 'relu':
        x = x.abs()
    if kind in ('wide_weights', 'dc50_wide'):       # magnitudes log-uniform over 10^-1.5 ... 10^1.5 around the init scale
        g = torch.Generator().manual_seed(seed + 2)
        wt = wt * torch.pow(10.0, torch.rand(shape_w, generator=g) * 3.0 - 1.5)
    return x, wt


def _winograd_stress(kind, k, pad, n, cin, cout, H, W, budget, info0, seed, wino1d4=0):
    import ctypes as C
    x, wt = _stress_operands(kind, (n, cin, H, W), (cout, cin, *k), seed)
    b = rnd((cout,), seed + 3, 0.1)
    want = F.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    scale = F.conv2d(x.double().abs(), wt.double().abs(), b.double().abs(), padding=pad)     # sum |w||x| (+ |b|)
    eps = 2.0 ** -24
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=pad)
    assert (pc.wwino if k == (3, 3) else pc.wwino1d) is not None
    lib = ops._lib.load()
    xd = x.to(DEV)
    prev = ops.set_conv_winograd(False)
    try:
        with ops.record_conv_kernels() as ran_d:
            direct = ops.conv2d(pc, xd)
        ops.set_conv_winograd(True)
        ops.tune('wino1d4', wino1d4)
        with ops.record_conv_kernels() as ran_w:
            got = ops.conv2d(pc, xd)
    finally:
        ops.tune('wino1d4', 1)
        ops.set_conv_winograd(prev)
    assert [kk for _, kk in ran_d] == ['direct-dma'], ran_d
    assert len(ran_w) == 1 and ran_w[0][1].startswith(info0), ran_w       # the Winograd kernel really ran
    r_dir = float(((direct.cpu().double() - want).abs() / (eps * scale)).max())
    r_win = float(((got.cpu().double() - want).abs() / (eps * scale)).max())
    r_rel = float(((got.cpu().double() - direct.cpu().double()).abs() / (eps * scale)).max())
    print(f'[measured] {info0} {kind} {cin}->{cout} {k} @{H}x{W}: err / (eps sum|w||x|) = {r_win:.2f} '
          f'(direct {r_dir:.2f}, winograd vs direct {r_rel:.2f}); max |out| {float(want.abs().max()):.1f}, '
          f'max abs err {float((got.cpu().double() - want).abs().max()):.2e}')
    assert r_dir <= 24.0, f'direct kernel {r_dir}'
    assert r_win <= budget, f'{info0} on {kind}: {r_win} eps sum|w||x| > budget {budget}'
    assert r_rel <= budget + 24.0       # |winograd - direct| <= both errors


@pytest.mark.parametrize('kind', WINO_STRESS_OPERANDS)
@pytest.mark.parametrize('shape', [(4, 128, 512, 32, 32), (2, 64, 64, 128, 128), (8, 256, 192, 32, 32), (8, 96, 96, 64, 64)])
def test_conv2d_winograd_stress_operands(kind, shape):
    """F(2x2, 3x3) vs torch fp64 AND vs the direct kernel on DC-offset / one-signed / wide-range operands,
    tolerance relative to sum |w||x| (see above).  96 -> 96 has an odd fragment count: the pair kernel; the other
    shapes run the quarter-domain kernel."""
    n, cin, cout, H, W = shape
    _winograd_stress(kind, (3, 3), 1, n, cin, cout, H, W, budget=12.0, info0='winograd', seed=700 + cin)


@pytest.mark.parametrize('kind', WINO_STRESS_OPERANDS)
@pytest.mark.parametrize('shape', [(8, 256, 256, (1, 5), 32, 32), (8, 256, 128, (5, 1), 32, 32), (2, 256, 256, (5, 1), 60, 80)])
def test_conv2d_winograd_1d_stress_operands(kind, shape):
    """F(2, 5) (points 0, +-1, +-2, inf: the worse-conditioned of the two) on the same stress operands."""
    n, cin, cout, k, H, W = shape
    pad = (0, 2) if k == (1, 5) else (2, 0)
    _winograd_stress(kind, k, pad, n, cin, cout, H, W, budget=60.0, info0='winograd F(2,5)', seed=800 + cout)


@pytest.mark.parametrize('kind', WINO_STRESS_OPERANDS)
@pytest.mark.parametrize('shape', [(8, 256, 256, (1, 5), 32, 32), (8, 256, 128, (5, 1), 32, 32), (2, 256, 256, (5, 1), 60, 80)])
def test_conv2d_winograd_1d4_stress_operands(kind, shape):
    """F(4, 5) (points 0, +-1, +-2, +-1/2, inf) on the same stress operands and shapes, same budget as F(2, 5)."""
    n, cin, cout, k, H, W = shape
    pad = (0, 2) if k == (1, 5) else (2, 0)
    _winograd_stress(kind, k, pad, n, cin, cout, H, W, budget=60.0, info0='winograd F(4,5)', seed=800 + cout, wino1d4=2)


@pytest.mark.parametrize('form', ['F(2,5)', 'F(4,5)'])
def test_sepconv_gru_winograd_drift_12_iterations(form):
    """configs[4]'s recurrence: 12 iterations of the SepConvGRU at (8, 60, 80) on the F(2, 5) / the F(4, 5) kernel vs the direct
    kernels from the same state, fresh motion features every iteration, post-ReLU (one-signed) context and motion
    channels like the real network's.  The gates contract (|dh'| <= max(z, 1 - z) |dh| + ...), so the difference
    must stay at round-off level instead of growing with the iteration count."""
    from scflow_amd.modules import ConvGRU
    torch.manual_seed(12)
    n, h, w = 8, 60, 80
    hc, cc, xc = 128, 128, 128
    gru = ConvGRU(hc, cc + xc, 'SeqConv').to(DEV)
    for prm in gru.parameters():
        prm.data.mul_(1.5)
    hx = rnd((n, hc + cc + xc, h, w), 195)
    hx[:, :hc] = torch.tanh(hx[:, :hc])
    hx[:, hc:] = torch.relu(hx[:, hc:] + 0.5)                      # context | motion features: post-ReLU, DC offset
    hist = {}
    for wino in (True, False):
        prev = ops.set_conv_winograd(wino)
        ops.tune('wino1d4', 1 if form == 'F(4,5)' else 0)     # 1: the dispatch's own choice -- F(4, 5) on all 4 launches here
        try:
            gru.invalidate_packed()
            a = hx.to(DEV)
            ctx = gru.context_terms(a[:, hc:hc + cc])
            states = []
            with ops.record_conv_kernels() as ran:
                for it in range(12):
                    a[:, hc + cc:] = torch.relu(rnd((n, xc, h, w), 196 + it) + 0.5).to(DEV)
                    gru.forward_inplace(a, ctx, cc)
                    states.append(a[:, :hc].clone())
            want_kind = 'winograd ' + form if wino else 'direct-dma'
            assert len(ran) == 48 and all(k == want_kind for _, k in ran), ran[:4]
            hist[wino] = states
        finally:
            ops.tune('wino1d4', 1)
            ops.set_conv_winograd(prev)
    errs = [float((a_ - b_).abs().max()) for a_, b_ in zip(hist[True], hist[False])]
    print(f'[measured] SepConvGRU {form} vs direct, (8, 60, 80), max |dh| per iteration: ' + ' '.join(f'{e:.1e}' for e in errs))
    assert errs[0] > 0.0
    assert max(errs) <= 1.2e-5, errs          # measured 3.0e-6 ... 3.8e-6 in every one of the 12 iterations
    assert errs[-1] <= 4.0 * max(errs[:3]) + 1e-6, f'the difference grows with the iteration count: {errs}'


def test_conv2d_dma_two_segments_gru_q():
    """the GRU candidate conv on the DMA kernel: two input segments, tanh gate epilogue."""
    n, h, w = 32, 32, 32
    hh, xx = rnd((n, 128, h, w), 40), rnd((n, 256, h, w), 41)
    z = torch.sigmoid(rnd((n, 128, h, w), 42))
    wt = rnd((128, 384, 1, 5), 43, 0.02)
    b = rnd((128,), 44, 0.1)
    rh = torch.sigmoid(rnd((n, 128, h, w), 45)) * hh
    q = torch.tanh(F.conv2d(torch.cat([rh, xx], 1), wt, b, padding=(0, 2)))
    want = (1 - z) * hh + z * q
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=(0, 2))
    out = torch.empty((n, 128, h, w), device=DEV)
    ops.conv2d(pc, rh.to(DEV), xx.to(DEV), out=out, mode=ops.CONV_GRU_Q, gru_h=hh.to(DEV),
               gru_z=z.to(DEV))
    close(out, want, atol=3e-5, what='gru q on dma kernel')


@pytest.mark.parametrize('cout,k', [(2, 3), (1, 1), (1, 3), (4, 3), (3, 1)])
def test_conv2d_thin_output(cout, k):
    """Cout <= 4 layers (flow / mask prediction) run on the vector-ALU kernel."""
    n, cin, H, W = 3, 256, 32, 32
    x = rnd((n, cin + 64, H, W), 50)
    wt = rnd((cout, cin, k, k), 51, (1.0 / (cin * k * k)) ** 0.5)
    b = rnd((cout,), 52, 0.1)
    for act, fn in ((ops.ACT_NONE, lambda t: t), (ops.ACT_SIGMOID, torch.sigmoid)):
        want = fn(F.conv2d(x[:, 64:], wt, b, padding=k // 2))
        pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=k // 2)
        got = ops.conv2d(pc, x.to(DEV)[:, 64:], act=act)        # channel-slice view as input
        close(got, want, atol=2e-5, what=f'thin {cout} {k} {act}')
    xs = rnd((2, cin, 12, 20), 53)                              # ragged tile
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=k // 2)
    close(ops.conv2d(pc, xs.to(DEV)), F.conv2d(xs, wt, b, padding=k // 2), atol=2e-5, what='thin ragged')
    # maps wider than one 32-column tile (the 3x3 kernel takes a tile's edge columns from memory, the inner ones from
    # the neighbouring lanes): 60 x 80 (configs[4]), an odd width with a ragged last tile, one column
    for hh, ww in ((60, 80), (9, 67), (5, 33), (4, 1)):
        xw = rnd((2, cin, hh, ww), 54 + ww)
        close(ops.conv2d(pc, xw.to(DEV)), F.conv2d(xw, wt, b, padding=k // 2), atol=2e-5, what=f'thin {hh}x{ww}')


def test_conv2d_two_segments_into_channel_slice():
    n, h, w = 2, 16, 16
    xa, xb = rnd((n, 192, h, w), 20), rnd((n, 64, h, w), 21)
    wt = rnd((126, 256, 3, 3), 22, 0.02)
    b = rnd((126,), 23, 0.1)
    want = torch.relu(F.conv2d(torch.cat([xa, xb], 1), wt, b, padding=1))
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=1)
    big = torch.full((n, 256, h, w), -7.0, device=DEV)
    buf_a = torch.zeros((n, 200, h, w), device=DEV)
    buf_a[:, 8:200] = xa.to(DEV)
    ops.conv2d(pc, buf_a[:, 8:200], xb.to(DEV), out=big[:, 128:254], act=ops.ACT_RELU)
    close(big[:, 128:254], want, atol=2e-5, what='slice')
    assert float((big[:, :128] + 7.0).abs().max()) == 0.0 and float((big[:, 254:] + 7.0).abs().max()) == 0.0


def test_conv2d_bn_residual_split_act():
    n, c, h, w = 2, 64, 16, 16
    x, res = rnd((n, c, h, w), 30), rnd((n, 96, h, w), 31)
    wt, b = rnd((96, c, 3, 3), 32, 0.05), rnd((96,), 33, 0.1)
    gamma, beta = 1 + 0.1 * rnd((96,), 34), 0.1 * rnd((96,), 35)
    mean, var = 0.1 * rnd((96,), 36), 1 + 0.2 * rnd((96,), 37).abs()
    y = F.batch_norm(F.conv2d(x, wt, b, padding=1), mean, var, gamma, beta, False, 0., 1e-5)
    want = torch.relu(y + res)
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=1,
                                    bn=[t.to(DEV) for t in (gamma, beta, mean, var)])
    got = ops.conv2d(pc, x.to(DEV), res=res.to(DEV), act=ops.ACT_RELU)
    close(got, want, atol=3e-5, what='bn+res')
    # split activation: tanh on the first 32 channels, relu on the rest
    pc2 = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=1)
    y2 = F.conv2d(x, wt, b, padding=1)
    want2 = torch.cat([torch.tanh(y2[:, :32]), torch.relu(y2[:, 32:])], 1)
    got2 = ops.conv2d(pc2, x.to(DEV), act=ops.ACT_TANH, act2=ops.ACT_RELU, act_split=32)
    close(got2, want2, atol=2e-5, what='split act')


def test_conv2d_gru_fusions():
    """SepConvGRU pass (raft_decoder.py:235-253) via the fused ZR / Q epilogues."""
    n, h, w = 2, 8, 8
    hx = rnd((n, 384, h, w), 40)
    hx[:, :128] = torch.tanh(hx[:, :128])
    wz, wr, wq = (rnd((128, 384, 1, 5), s, 0.03) for s in (41, 42, 43))
    bz, br, bq = (rnd((128,), s, 0.1) for s in (44, 45, 46))
    hcur, x = hx[:, :128], hx[:, 128:]
    z = torch.sigmoid(F.conv2d(hx, wz, bz, padding=(0, 2)))
    r = torch.sigmoid(F.conv2d(hx, wr, br, padding=(0, 2)))
    q = torch.tanh(F.conv2d(torch.cat([r * hcur, x], 1), wq, bq, padding=(0, 2)))
    want = (1 - z) * hcur + z * q
    pzr = ops.PackedConv.from_weight(torch.cat([wz, wr]).to(DEV), torch.cat([bz, br]).to(DEV),
                                     padding=(0, 2))
    pq = ops.PackedConv.from_weight(wq.to(DEV), bq.to(DEV), padding=(0, 2))
    hxd = hx.to(DEV)
    rh = torch.empty((n, 128, h, w), device=DEV)
    zr_out = torch.empty((n, 128, h, w), device=DEV)
    ops.conv2d(pzr, hxd, out=zr_out, mode=ops.CONV_GRU_ZR, gru_h=hxd[:, :128], gru_aux=rh)
    close(zr_out, z, atol=2e-5, what='z')
    close(rh, r * hcur, atol=2e-5, what='r*h')
    ops.conv2d(pq, rh, hxd[:, 128:], out=hxd[:, :128], mode=ops.CONV_GRU_Q, gru_h=hxd[:, :128],
               gru_z=zr_out)
    close(hxd[:, :128], want, atol=3e-5, what='h_new')
    close(hxd[:, 128:], x, atol=0, what='x untouched')


@pytest.mark.parametrize('n,h,w,kind', [(2, 8, 8, 'SeqConv'), (32, 32, 32, 'SeqConv'), (1, 32, 32, 'SeqConv'),
                                         (1, 60, 80, 'SeqConv'), (2, 12, 20, 'Conv')])
def test_convgru_context_hoisting(n, h, w, kind):
    """ConvGRU with the context channels' part of the convolutions evaluated once
    (scf_sepconv_gru_ctx) == the plain cell (scf_sepconv_gru) and == torch
    (raft_decoder.py:235-253) within fp32 round-off, over several iterations with the same
    context and changing motion features."""
    from scflow_amd.modules import ConvGRU
    torch.manual_seed(11)
    hc, cc, xc = 128, 128, 128
    gru = ConvGRU(hc, cc + xc, kind)
    for prm in gru.parameters():
        prm.data.mul_(1.5)
    gru_d = ConvGRU(hc, cc + xc, kind).to(DEV)
    gru_d.load_state_dict(gru.state_dict())
    hx = rnd((n, hc + cc + xc, h, w), 90)
    hx[:, :hc] = torch.tanh(hx[:, :hc])
    a, b = hx.to(DEV), hx.to(DEV)
    ctx = gru_d.context_terms(b[:, hc:hc + cc])
    assert len(ctx) == len(gru.conv_z) and ctx[0].shape == (n, 3 * hc, h, w)
    href = hx[:, :hc].clone()
    for it in range(3):
        mot = rnd((n, xc, h, w), 91 + it)
        a[:, hc + cc:] = mot.to(DEV)
        b[:, hc + cc:] = mot.to(DEV)
        gru_d.forward_inplace(a)
        gru_d.forward_inplace(b, ctx, cc)
        x = torch.cat([hx[:, hc:hc + cc], mot], 1)
        for cz, cr, cq in zip(gru.conv_z, gru.conv_r, gru.conv_q):
            hxr = torch.cat([href, x], 1)
            z = torch.sigmoid(F.conv2d(hxr, cz.conv.weight, cz.conv.bias, padding=cz.conv.padding))
            r = torch.sigmoid(F.conv2d(hxr, cr.conv.weight, cr.conv.bias, padding=cr.conv.padding))
            q = torch.tanh(F.conv2d(torch.cat([r * href, x], 1), cq.conv.weight, cq.conv.bias,
                                    padding=cq.conv.padding))
            href = ((1 - z) * href + z * q).detach()
        close(b[:, :hc], href, atol=5e-5, what=f'hoisted vs torch, iteration {it}')
        close(b[:, :hc], a[:, :hc].cpu(), atol=2e-5, what=f'hoisted vs plain cell, iteration {it}')
        close(b[:, hc:], a[:, hc:].cpu(), atol=0, what='x untouched')


@pytest.mark.parametrize('n,h,w', [(32, 32, 32), (8, 60, 80), (6, 21, 28)])
def test_wino1d4_half_domain_kernel(n, h, w):
    """The half-domain F(4, 5) kernel (conv_wino1d4h_kernel, ``ops.tune('wino1d4_half', 1)``: a wave holds 4 of the 8
    transform positions for two channel fragments, one exchange of the output shares per block; off by default -- measured
    no faster in the step) against the full-domain kernel: plain 1x5 / 5x1 layers with two input segments, then the GRU
    cell (both gate epilogues, the context term streamed through the accumulators of the position-0/1/2/7 waves for both
    fragments).  Same B operands, same accumulators; the outputs differ by the association of the last adds."""
    from scflow_amd.modules import ConvGRU
    try:
        ops.tune('wino1d4', 2)
        for k, pad in (((1, 5), (0, 2)), ((5, 1), (2, 0))):
            cin, cout, c0 = 104, 128, 40
            x = rnd((n, cin, h, w), 310).to(DEV)
            wt = rnd((cout, cin, *k), 311, (1.0 / (cin * 5)) ** 0.5)
            b = rnd((cout,), 312, 0.1)
            pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=pad)
            outs = []
            for hv in (0, 1):
                ops.tune('wino1d4_half', hv)
                with ops.record_conv_kernels() as ran:
                    outs.append(ops.conv2d(pc, x[:, :c0], x[:, c0:], act=ops.ACT_RELU))
                assert ran[0][1] == 'winograd F(4,5)', ran
            want = torch.relu(F.conv2d(x.cpu().double(), wt.double(), b.double(), padding=pad)).float()
            close(outs[1], want, atol=3e-5, what=f'half-domain {k}')
            d = float((outs[0] - outs[1]).abs().max())
            assert d <= 1e-5, d
        torch.manual_seed(12)
        hc, cc, xc = 128, 128, 128
        gru = ConvGRU(hc, cc + xc, 'SeqConv').to(DEV)
        hx = rnd((n, hc + cc + xc, h, w), 95)
        hx[:, :hc] = torch.tanh(hx[:, :hc])
        states = []
        for hv in (0, 1):
            ops.tune('wino1d4_half', hv)
            gru.invalidate_packed()
            a = hx.to(DEV)
            ctx = gru.context_terms(a[:, hc:hc + cc])
            with ops.record_conv_kernels() as ran:
                for it in range(3):
                    gru.forward_inplace(a, ctx, cc)
            assert len(ran) == 12 and all(kk == 'winograd F(4,5)' for _, kk in ran), ran
            states.append(a[:, :hc].clone())
        d = float((states[0] - states[1]).abs().max())
        print(f'[measured] F(4,5) half- vs full-domain kernel, GRU state after 3 iterations ({n}, {h}, {w}): max |dh| {d:.1e}')
        assert 0.0 < d <= 1e-5, d
    finally:
        ops.tune('wino1d4_half', 0)
        ops.tune('wino1d4', 1)


@pytest.mark.parametrize('form', ['F(2,5)', 'F(4,5)'])
@pytest.mark.parametrize('n,h,w', [(32, 32, 32), (8, 60, 80)])
def test_sepconv_gru_winograd(n, h, w, form):
    """SepConvGRU with its 1x5 / 5x1 gates on the F(2, 5) kernel (conv_wino1d.hip: both GRU epilogues, two input
    segments, hoisted context term) or, as the dispatch chooses at these sizes, the F(4, 5) kernel (conv_wino1d4.hip:
    the context term enters through the accumulators) vs the direct kernels, two iterations: the state differs by
    fp32 round-off."""
    from scflow_amd.modules import ConvGRU
    torch.manual_seed(12)
    hc, cc, xc = 128, 128, 128
    gru = ConvGRU(hc, cc + xc, 'SeqConv').to(DEV)
    for prm in gru.parameters():
        prm.data.mul_(1.5)
    hx = rnd((n, hc + cc + xc, h, w), 95)
    hx[:, :hc] = torch.tanh(hx[:, :hc])
    outs = {}
    for wino in (True, False):
        prev = ops.set_conv_winograd(wino)
        ops.tune('wino1d4', 1 if form == 'F(4,5)' else 0)
        try:
            gru.invalidate_packed() if hasattr(gru, 'invalidate_packed') else None
            a = hx.to(DEV)
            ctx = gru.context_terms(a[:, hc:hc + cc])
            with ops.record_conv_kernels() as ran:
                for it in range(2):
                    a[:, hc + cc:] = rnd((n, xc, h, w), 96 + it).to(DEV)
                    gru.forward_inplace(a, ctx, cc)
            assert len(ran) == 8 and all(k == ('winograd ' + form if wino else 'direct-dma') for _, k in ran), ran
            outs[wino] = a[:, :hc].clone()
        finally:
            ops.tune('wino1d4', 1)
            ops.set_conv_winograd(prev)
    err = float((outs[True] - outs[False]).abs().max())
    print(f'[measured] SepConvGRU {form} vs direct kernels, {n}x{h}x{w}: max |dh| after 2 iterations {err:.2e}')
    assert 0.0 < err <= 2e-5, err


# ----------------------------------------------------------- small kernels
@pytest.mark.parametrize('hw', [(128, 128), (64, 64), (32, 32), (12, 20), (5, 7), (240, 320), (120, 160), (250, 330)])
def test_instance_norm(hw):
    x, res = rnd((2, 6, *hw), 50, 2.0) + 0.5, rnd((2, 6, *hw), 51)
    close(ops.instance_norm(x.to(DEV), relu=True), torch.relu(F.instance_norm(x, eps=1e-5)),
          atol=2e-5, what='in+relu')
    close(ops.instance_norm(x.to(DEV)), F.instance_norm(x, eps=1e-5), atol=2e-5, what='in')
    close(ops.instance_norm(x.to(DEV), res=res.to(DEV), relu=True),
          torch.relu(F.instance_norm(x, eps=1e-5) + res), atol=2e-5, what='in+res+relu')


@pytest.mark.parametrize('hw', [(16, 16), (8, 8), (4, 4), (32, 32), (5, 7), (23, 23)])      # register-resident groups, and larger ones
def test_group_norm_relu(hw):
    x = rnd((3, 128, *hw), 52, 1.5)
    g, b = 1 + 0.1 * rnd((128,), 53), 0.1 * rnd((128,), 54)
    close(ops.group_norm_relu(x.to(DEV), g.to(DEV), b.to(DEV), 32),
          torch.relu(F.group_norm(x, 32, g, b, 1e-5)), atol=2e-5, what='gn')


@pytest.mark.parametrize('n,k,o', [(3, 2048, 1024), (1, 1024, 256), (32, 256, 126), (5, 30, 7), (32, 2048, 1024),
                                   (40, 4096, 70), (33, 72, 33), (2, 60, 5)])
def test_linear(n, k, o):
    x, wt, b = rnd((n, k), 55), rnd((o, k), 56, k ** -0.5), rnd((o,), 57, 0.1)
    close(ops.linear(x.to(DEV), wt.to(DEV), b.to(DEV), ops.ACT_RELU), torch.relu(F.linear(x, wt, b)),
          atol=2e-5, what='linear')


@pytest.mark.parametrize('n,k,o,slices', [(32, 2048, 1024, 8), (3, 2048, 1024, 8), (1, 1024, 256, 4), (40, 512, 70, 2),
                                          (32, 256, 126, 1), (5, 64, 33, 1), (33, 1024, 256, 4)])
def test_fc_splitk_vs_torch(n, k, o, slices):
    """scf_fc_splitk (split-K MFMA GEMM): finished outputs (slices = 1) and partial sums whose slice-ordered sum
    + bias + ReLU -- what the next layer's operand load computes -- equals nn.Linear + ReLU"""
    x, wt, b = rnd((n, k), 155), rnd((o, k), 156, k ** -0.5), rnd((o,), 157, 0.1)
    want = torch.relu(F.linear(x.double(), wt.double(), b.double())).float()
    if slices == 1:
        got = ops.fc_splitk(x.to(DEV), wt.to(DEV), b.to(DEV), act=ops.ACT_RELU)
        close(got, want, atol=2e-6, what='fc finished')
    else:
        parts = ops.fc_splitk(x.to(DEV), wt.to(DEV), slices=slices)
        assert parts.shape == (slices, n, o)
        got = torch.relu(parts.sum(0) + b.to(DEV))
        close(got, want, atol=2e-6, what='fc partial sums')
        # consumed by a next layer: identity weights read the summed, biased, rectified features back
        if o % 8 == 0 and o <= 256:
            eye = torch.eye(o, device=DEV)
            back = ops.fc_splitk(parts, eye, x_bias=b.to(DEV), x_relu=True)
            close(back, want, atol=2e-6, what='fc partials through the next load')
    assert ops.fc_slices(k) == slices or slices in (1, 2)


def test_fc_splitk_group_norm_and_two_heads():
    """the pose head's tail as three launches: GroupNorm(32) + ReLU folded into fc1's load, bias + ReLU of fc1 / fc2
    folded into the next loads, rotation | translation heads in one launch -- against torch, and the rejected shapes"""
    from scflow_amd._lib import ScflowHipError
    n = 32
    y3 = rnd((n, 128, 4, 4), 160, 1.5) + 0.3
    gamma, beta = 1 + 0.1 * rnd((128,), 161), 0.1 * rnd((128,), 162)
    w1, b1 = rnd((1024, 2048), 163, 2048 ** -0.5), rnd((1024,), 164, 0.1)
    w2, b2 = rnd((256, 1024), 165, 1024 ** -0.5), rnd((256,), 166, 0.1)
    wr, br, wt_, bt = rnd((126, 256), 167, 0.06), rnd((126,), 168, 0.1), rnd((63, 256), 169, 0.06), rnd((63,), 170, 0.1)
    f = torch.relu(F.group_norm(y3.double(), 32, gamma.double(), beta.double(), 1e-5)).flatten(1)
    h1 = torch.relu(F.linear(f, w1.double(), b1.double()))
    h2 = torch.relu(F.linear(h1, w2.double(), b2.double()))
    want_r, want_t = F.linear(h2, wr.double(), br.double()).float(), F.linear(h2, wt_.double(), bt.double()).float()
    D = lambda t: t.to(DEV)
    p1 = ops.fc_splitk(D(y3).view(n, -1), D(w1), gn=(32, 16, D(gamma), D(beta), 1e-5), slices=8)
    close(torch.relu(p1.sum(0) + D(b1)), h1.float(), atol=5e-6, what='gn + fc1')
    p2 = ops.fc_splitk(p1, D(w2), x_bias=D(b1), x_relu=True, slices=4)
    got_r, got_t = ops.fc_splitk(p2, D(wr), D(br), x_bias=D(b2), x_relu=True, weight2=D(wt_), bias2=D(bt))
    close(got_r, want_r, atol=3e-6, what='rotation head')
    close(got_t, want_t, atol=3e-6, what='translation head')
    # deterministic: the same bits run after run
    again = ops.fc_splitk(p2, D(wr), D(br), x_bias=D(b2), x_relu=True, weight2=D(wt_), bias2=D(bt))
    assert torch.equal(again[0], got_r) and torch.equal(again[1], got_t)
    with pytest.raises(ScflowHipError):          # a K-slice of 300 features is not a multiple of 8
        ops.fc_splitk(D(rnd((4, 600), 1)), D(rnd((8, 600), 2)), slices=2)
    with pytest.raises(ScflowHipError):          # groups of 48 features do not tile a 256-feature slice
        ops.fc_splitk(D(rnd((4, 768), 1)), D(rnd((8, 768), 2)), gn=(16, 16, D(gamma), D(beta), 1e-5), slices=3)


def test_pose_head_fused_tail_vs_linear_launches():
    """MultiClassPoseHead.features with the fused tail (default) against the GroupNorm + scf_linear launches"""
    from scflow_amd.registry import HEAD, build_from_cfg
    import scflow_amd
    head = build_from_cfg(scflow_amd.scflow_model_cfg()['decoder']['pose_head_cfg'], HEAD)
    torch.manual_seed(5)
    for prm in head.parameters():
        prm.data.copy_(torch.randn_like(prm) * (0.05 if prm.dim() > 1 else 0.1))
    head = head.to(DEV)
    for n in (1, 3, 32):
        x = rnd((n, 224, 32, 32), 180 + n).to(DEV)
        assert head.fc_plan() == (8, 4)
        with ops.record_conv_kernels() as ran:
            r1, t1 = head.features(x)
        head.fused_fc = False
        r0, t0 = head.features(x)
        head.fused_fc = True
        assert len(ran) == 3
        close(r1, r0.cpu(), atol=2e-6, what=f'rotation rows, N={n}')
        close(t1, t0.cpu(), atol=2e-6, what=f'translation rows, N={n}')


@pytest.mark.parametrize('n,k', [(1, 256), (3, 256), (32, 250), (2, 2048)])
def test_linear_pair(n, k):
    """scf_linear_pair (rotation_pred + translation_pred, pose_head.py:203-206) == two scf_linear
    calls bit for bit, == torch within round-off."""
    x = rnd((n, k), 58)
    w1, b1, w2, b2 = rnd((126, k), 59, k ** -0.5), rnd((126,), 60, 0.1), rnd((63, k), 61, k ** -0.5), None
    y1, y2 = ops.linear_pair(x.to(DEV), w1.to(DEV), b1.to(DEV), w2.to(DEV), b2)
    close(y1, F.linear(x, w1, b1), atol=2e-5, what='pair first')
    close(y2, F.linear(x, w2), atol=2e-5, what='pair second')
    assert torch.equal(y1, ops.linear(x.to(DEV), w1.to(DEV), b1.to(DEV)))
    assert torch.equal(y2, ops.linear(x.to(DEV), w2.to(DEV), None))


def test_pose_update_and_label_quirk(golden_dir):
    g = np.load(os.path.join(golden_dir, 'pose_math.npz'))
    n, nc = 3, 21
    rot_all, tr_all = rnd((n, nc * 6), 60), rnd((n, nc * 3), 61, 0.05)
    label = torch.tensor([2, 5, 7])
    d_rot, d_tr = torch.from_numpy(g['d_rot']), torch.from_numpy(g['d_trans'])
    for i in range(n):                                  # plant the golden deltas at class label[0]
        rot_all.view(n, nc, 6)[i, 2] = d_rot[i]
        tr_all.view(n, nc, 3)[i, 2] = d_tr[i]
    R, t = torch.from_numpy(g['rot']), torch.from_numpy(g['trans'])
    o = ops.pose_update(rot_all.to(DEV), tr_all.to(DEV), label.to(DEV), nc, R.to(DEV), t.to(DEV), 0)
    close(o[0], d_rot, atol=0, what='d_rot select (reference quirk: label[0] for all)')
    close(o[2], torch.from_numpy(g['rot_new']), atol=2e-6, what='R')
    close(o[3], torch.from_numpy(g['trans_new']), atol=2e-4, what='t')
    o1 = ops.pose_update(rot_all.to(DEV), tr_all.to(DEV), label.to(DEV), nc, R.to(DEV), t.to(DEV), 1)
    want = rot_all.view(n, nc, 6)[torch.arange(n), label]
    close(o1[0], want, atol=0, what='per-sample label mode')


def test_pose_update_linear_depth_transform(golden_dir):
    """pose.py:139-141 (depth_transform other than 'exp'): SCF_POSE_DEPTH_LINEAR, alone and with the per-sample label
    bit; a bit outside the set is refused."""
    g = np.load(os.path.join(golden_dir, 'pose_math_linear.npz'))
    n, nc = 3, 21
    rot_all, tr_all = rnd((n, nc * 6), 60), rnd((n, nc * 3), 61, 0.05)
    label = torch.tensor([2, 5, 7])
    d_rot, d_tr = torch.from_numpy(g['d_rot']), torch.from_numpy(g['d_trans'])
    for i in range(n):
        rot_all.view(n, nc, 6)[i, 2] = d_rot[i]                 # label_mode bit 0 clear: class label[0] = 2 for all
        tr_all.view(n, nc, 3)[i, 2] = d_tr[i]
    R, t = torch.from_numpy(g['rot']), torch.from_numpy(g['trans'])
    a = lambda *x: [v.to(DEV) for v in x]
    o = ops.pose_update(*a(rot_all, tr_all, label), nc, *a(R, t), 2)
    close(o[2], torch.from_numpy(g['rot_new']), atol=2e-6, what='R (linear)')
    close(o[3], torch.from_numpy(g['trans_new']), atol=2e-4, what='t (linear)')
    o_exp = ops.pose_update(*a(rot_all, tr_all, label), nc, *a(R, t), 0)
    assert float((o_exp[3] - o[3]).abs().max()) > 1e-2
    # both bits: sample i decoded with class label[i], linear depth
    for i in range(n):
        rot_all.view(n, nc, 6)[i, int(label[i])] = d_rot[i]
        tr_all.view(n, nc, 3)[i, int(label[i])] = d_tr[i]
        if i:
            tr_all.view(n, nc, 3)[i, 2] = 9.0                   # poison what label[0] would select
    o3 = ops.pose_update(*a(rot_all, tr_all, label), nc, *a(R, t), 3)
    close(o3[3], torch.from_numpy(g['trans_new']), atol=2e-4, what='t (linear, per-sample label)')
    with pytest.raises(Exception):
        ops.pose_update(*a(rot_all, tr_all, label), nc, *a(R, t), 4)


def test_reproject_and_unproject(golden_dir):
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, 'pose_math.npz')).items()
         if v.dtype.kind == 'f'}
    d = lambda k: g[k].to(DEV)
    for inv, key in ((0., 'flow_inv0'), (400., 'flow_inv400')):
        got = ops.reproject_flow(d('depth'), d('k'), d('rot'), d('trans'), d('rot_new'),
                                 d('trans_new'), inv)
        close(got, g[key], atol=2e-4, what=key)
    pts = ops.unproject_depth(d('depth'), d('k'), d('rot'), d('trans')).cpu()
    p2, p3 = oracle.unproject_depth(g['depth'][0], g['k'][0], g['rot'][0], g['trans'][0])
    dense = pts[0][:, p2[:, 1].long(), p2[:, 0].long()].t()
    close(dense, p3, atol=2e-3, what='pts3d')


def test_reproject_full_size_matches_oracle():
    inp = make_inputs(2, 256, 256, seed=3)
    r, t = inp['ref_rotation'], inp['ref_translation'] + torch.tensor([1.5, -2.0, 10.0])
    pts = [oracle.unproject_depth(inp['depth'][i], inp['internel_k'][i], inp['ref_rotation'][i],
                                  inp['ref_translation'][i]) for i in range(2)]
    want = oracle.flow_from_pose_and_points(r, t, inp['internel_k'], [a for a, _ in pts],
                                            [b for _, b in pts], 256, 256, 0.)
    got = ops.reproject_flow(inp['depth'].to(DEV), inp['internel_k'].to(DEV), r.to(DEV),
                             inp['ref_translation'].to(DEV), r.to(DEV), t.to(DEV), 0.)
    close(got, want, atol=5e-4, what='flow 256')


@pytest.mark.parametrize('shape,out', [((2, 2, 256, 256), (32, 32)), ((2, 2, 32, 32), (256, 256)),
                                       ((1, 1, 32, 32), (256, 256)), ((1, 3, 12, 20), (5, 9)), ((3, 2, 60, 80), (480, 640)),
                                       ((2, 1, 9, 7), (30, 301)), ((1, 2, 8, 8), (1, 1))])
def test_resize_bilinear(shape, out):
    a, b = rnd(shape, 70, 4.0), rnd(shape, 71)
    want = 0.125 * F.interpolate(a + b, size=out, mode='bilinear', align_corners=True)
    close(ops.resize_bilinear(a.to(DEV), out, 0.125, b.to(DEV)), want, atol=2e-6, what='resize')
    want = F.interpolate(a, size=out, mode='bilinear', align_corners=True)
    close(ops.resize_bilinear(a.to(DEV), out), want, atol=2e-6, what='resize')


def test_avgpool_and_copy():
    x = rnd((3, 5, 12, 20), 80)
    close(ops.avgpool2x2(x.to(DEV)), F.avg_pool2d(x, 2, 2), atol=1e-6, what='pool')
    dst = torch.zeros((3, 9, 12, 20), device=DEV)
    ops.copy_channels(x.to(DEV), dst[:, 2:7])
    close(dst[:, 2:7], x, atol=0, what='copy')


# ------------------------------------------------------- split-fp16 (3 x MFMA) convolution
F16_CASES = [c for c in CONV_CASES if c[1] >= 16 and c[3] != (1, 1)]
F16_CASES += [(2, 128, 128, (3, 3), 1, (1, 1), 60, 80),     # 16-column fragments (Wo = 80)
              (2, 64, 96, (5, 1), 1, (2, 0), 28, 40)]      # 8-column fragments (Wo = 40)


@pytest.mark.parametrize('case', F16_CASES)
def test_conv2d_f16x3(case):
    """fp32-class accuracy of the split-fp16 kernel: error vs fp64 must stay within a small
    multiple of the fp32 kernel's own error (both ~1e-6 relative to the output scale)."""
    n, cin, cout, k, s, p, H, W = case
    x = rnd((n, cin, H, W), 10)
    wt = rnd((cout, cin, *k), 11, (1.0 / (cin * k[0] * k[1])) ** 0.5)
    b = rnd((cout,), 12, 0.1)
    ref64 = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), stride=s, padding=p))
    pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), stride=s, padding=p)
    got32 = ops.conv2d(pc, x.to(DEV), act=ops.ACT_RELU).cpu().double()
    prev = ops.set_conv_precision('f16x3')
    try:
        got16 = ops.conv2d(pc, x.to(DEV), act=ops.ACT_RELU).cpu().double()
    finally:
        ops.set_conv_precision(prev)
    e32 = float((got32 - ref64).abs().max())
    e16 = float((got16 - ref64).abs().max())
    scale = float(ref64.abs().max())
    assert e16 <= 4e-6 * scale + 1e-6, f'{case}: f16x3 err {e16:.2e} (fp32 kernel {e32:.2e}, scale {scale:.2f})'


def test_conv2d_f16x3_gru_and_segments():
    n, h, w = 2, 16, 16
    hx = rnd((n, 384, h, w), 40)
    hx[:, :128] = torch.tanh(hx[:, :128])
    wz, wr, wq = (rnd((128, 384, 1, 5), s, 0.03) for s in (41, 42, 43))
    bz, br, bq = (rnd((128,), s, 0.1) for s in (44, 45, 46))
    hcur, x = hx[:, :128], hx[:, 128:]
    z = torch.sigmoid(F.conv2d(hx, wz, bz, padding=(0, 2)))
    r = torch.sigmoid(F.conv2d(hx, wr, br, padding=(0, 2)))
    q = torch.tanh(F.conv2d(torch.cat([r * hcur, x], 1), wq, bq, padding=(0, 2)))
    want = (1 - z) * hcur + z * q
    pzr = ops.PackedConv.from_weight(torch.cat([wz, wr]).to(DEV), torch.cat([bz, br]).to(DEV), padding=(0, 2))
    pq = ops.PackedConv.from_weight(wq.to(DEV), bq.to(DEV), padding=(0, 2))
    hxd = hx.to(DEV)
    zb = torch.empty((n, 128, h, w), device=DEV)
    rh = torch.empty((n, 128, h, w), device=DEV)
    prev = ops.set_conv_precision('f16x3')
    try:
        ops.conv2d(pzr, hxd, out=zb, mode=ops.CONV_GRU_ZR, gru_h=hxd[:, :128], gru_aux=rh)
        ops.conv2d(pq, rh, hxd[:, 128:], out=hxd[:, :128], mode=ops.CONV_GRU_Q, gru_h=hxd[:, :128], gru_z=zb)
    finally:
        ops.set_conv_precision(prev)
    close(zb, z, atol=2e-5, what='z')
    close(hxd[:, :128], want, atol=3e-5, what='h_new f16x3')


# ----------------------------------------------------------- whole GRU cell over the C ABI
@pytest.mark.parametrize('n,h,w,kind', [(2, 8, 8, 'SeqConv'), (32, 32, 32, 'SeqConv'), (1, 60, 80, 'SeqConv'),
                                         (2, 16, 16, 'Conv')])
def test_sepconv_gru_c_entry(n, h, w, kind):
    """scf_sepconv_gru (ConvGRU.forward, raft_decoder.py:235-253) with weights packed by the C
    host packers (what INTEGRATION.md's ctypes stub does) == the per-convolution launch sequence
    bit for bit, and == torch within fp32 round-off."""
    import ctypes as C
    from scflow_amd import _lib
    lib = _lib.load()
    ks = {'SeqConv': [((1, 5), (0, 2)), ((5, 1), (2, 0))], 'Conv': [((3, 3), (1, 1))]}[kind]
    ch, cx = 128, 256
    hx = rnd((n, ch + cx, h, w), 70)
    hx[:, :ch] = torch.tanh(hx[:, :ch])
    want_h, x = hx[:, :ch].clone(), hx[:, ch:]
    packs, keep = [], []
    passes = (_lib.GruPass * len(ks))()
    for i, (g, (k, pad)) in enumerate(zip(passes, ks)):
        wz, wr, wq = (rnd((ch, ch + cx, *k), 71 + 10 * i + s, 0.03) for s in range(3))
        bz, br, bq = (rnd((ch,), 74 + 10 * i + s, 0.1) for s in range(3))
        hxc = torch.cat([want_h, x], 1)
        z = torch.sigmoid(F.conv2d(hxc, wz, bz, padding=pad))
        r = torch.sigmoid(F.conv2d(hxc, wr, br, padding=pad))
        q = torch.tanh(F.conv2d(torch.cat([r * want_h, x], 1), wq, bq, padding=pad))
        want_h = (1 - z) * want_h + z * q
        wzr, bzr = torch.cat([wz, wr]).contiguous(), torch.cat([bz, br])
        packs.append((ops.PackedConv.from_weight(wzr.to(DEV), bzr.to(DEV), padding=pad),
                      ops.PackedConv.from_weight(wq.to(DEV), bq.to(DEV), padding=pad)))
        grp = 2 if k[0] * k[1] <= 5 else 1
        bufs = []
        for wt in (wzr, wq.contiguous()):
            co, ci, kh, kw = wt.shape
            a = torch.empty(lib.scf_pack_conv_weight_size(co, ci, kh, kw, 8))
            assert lib.scf_pack_conv_weight(wt.data_ptr(), co, ci, kh, kw, 8, a.data_ptr()) == 0
            b4 = torch.empty(lib.scf_pack_conv_weight_a4_size(co, ci, kh, kw, grp))
            assert lib.scf_pack_conv_weight_a4(wt.data_ptr(), co, ci, kh, kw, grp, b4.data_ptr()) == 0
            bufs += [a.to(DEV), b4.to(DEV)]
        bufs += [bzr.to(DEV), bq.to(DEV)]
        for wt in (wzr, wq.contiguous()):                 # KC = 32 packings (small-grid regime)
            co, ci, kh, kw = wt.shape
            a32 = torch.empty(lib.scf_pack_conv_weight_size(co, ci, kh, kw, 32))
            assert lib.scf_pack_conv_weight(wt.data_ptr(), co, ci, kh, kw, 32, a32.data_ptr()) == 0
            bufs.append(a32.to(DEV))
        g.wp_zr_k32, g.wp_q_k32 = bufs[6].data_ptr(), bufs[7].data_ptr()
        grps = ops.choose_a4s_groups(ch + cx, k[0], k[1], 1)      # small-grid a4 packings (bigger chunks)
        for wt in (wzr, wq.contiguous()):
            co, ci, kh, kw = wt.shape
            a4s = torch.empty(lib.scf_pack_conv_weight_a4_size(co, ci, kh, kw, grps))
            assert lib.scf_pack_conv_weight_a4(wt.data_ptr(), co, ci, kh, kw, grps, a4s.data_ptr()) == 0
            bufs.append(a4s.to(DEV))
        g.wp_zr_a4s, g.wp_q_a4s, g.a4s_groups = bufs[8].data_ptr(), bufs[9].data_ptr(), grps
        grpt = ops.choose_a4t_groups(ch + cx, k[0], k[1], 1)      # tiny-grid packings of 3x3 passes (32-channel chunks)
        if grpt:
            for wt in (wzr, wq.contiguous()):
                co, ci, kh, kw = wt.shape
                a4t = torch.empty(lib.scf_pack_conv_weight_a4_size(co, ci, kh, kw, grpt))
                assert lib.scf_pack_conv_weight_a4(wt.data_ptr(), co, ci, kh, kw, grpt, a4t.data_ptr()) == 0
                bufs.append(a4t.to(DEV))
            g.wp_zr_a4t, g.wp_q_a4t, g.a4t_groups = bufs[10].data_ptr(), bufs[11].data_ptr(), grpt
        if k in ((1, 5), (5, 1)):                     # F(2, 5) packings of the separable passes (large grids)
            w1 = []
            for wt in (wzr, wq.contiguous()):
                co, ci = wt.shape[:2]
                u = torch.empty(lib.scf_pack_conv_weight_wino1d_size(co, ci))
                taps = wt.reshape(co, ci, 5).contiguous()
                assert lib.scf_pack_conv_weight_wino1d(taps.data_ptr(), co, ci, u.data_ptr()) == 0
                w1.append(u.to(DEV))
            g.wp_zr_wino1d, g.wp_q_wino1d = w1[0].data_ptr(), w1[1].data_ptr()
            bufs += w1
            w4 = []                                   # and the F(4, 5) ones
            for wt in (wzr, wq.contiguous()):
                co, ci = wt.shape[:2]
                u = torch.empty(lib.scf_pack_conv_weight_wino1d4_size(co, ci))
                taps = wt.reshape(co, ci, 5).contiguous()
                assert lib.scf_pack_conv_weight_wino1d4(taps.data_ptr(), co, ci, u.data_ptr()) == 0
                w4.append(u.to(DEV))
            g.wp_zr_wino1d4, g.wp_q_wino1d4 = w4[0].data_ptr(), w4[1].data_ptr()
            bufs += w4
        keep += bufs
        g.KH, g.KW, g.pad_h, g.pad_w = k[0], k[1], pad[0], pad[1]
        g.wp_zr, g.wp_zr_a4, g.wp_q, g.wp_q_a4 = (t.data_ptr() for t in bufs[:4])
        g.bias_zr, g.bias_q, g.a4_groups = bufs[4].data_ptr(), bufs[5].data_ptr(), grp
    # (a) raw C entry with C-packed weights
    hxa = hx.to(DEV)
    za, rha = torch.empty((n, ch, h, w), device=DEV), torch.empty((n, ch, h, w), device=DEV)
    rc = lib.scf_sepconv_gru(hxa.data_ptr(), hxa.stride(0), n, ch, cx, h, w, passes, len(ks),
                             za.data_ptr(), rha.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    # (b) ops.sepconv_gru (the modules' path) and (c) the launch-by-launch path (timers armed)
    hxb, hxc_ = hx.to(DEV), hx.to(DEV)
    zb, rhb = torch.empty_like(za), torch.empty_like(za)
    ops.sepconv_gru(packs, hxb, ch, zb, rhb)
    ops.conv_timing(True)
    try:
        ops.sepconv_gru(packs, hxc_, ch, zb, rhb)
    finally:
        ops.conv_timing(False)
    da, db, dc = (float((t[:, :ch].cpu() - want_h).abs().max()) for t in (hxa, hxb, hxc_))
    assert torch.equal(hxa, hxb) and torch.equal(hxb, hxc_), \
        f'max |h - torch|: C entry/C packers {da:.2e}, ops.sepconv_gru {db:.2e}, launch by launch {dc:.2e}; ' \
        f'|b - c| {float((hxb - hxc_).abs().max()):.2e}'
    close(hxa[:, :ch], want_h, atol=5e-5, what='h_new')
    close(hxa[:, ch:], x, atol=0, what='x untouched')
    # argument checks
    assert lib.scf_sepconv_gru(None, 0, n, ch, cx, h, w, passes, 1, za.data_ptr(), rha.data_ptr(), None) < 0
    assert lib.scf_sepconv_gru(hxa.data_ptr(), hxa.stride(0), n, 100, cx, h, w, passes, 1, za.data_ptr(),
                               rha.data_ptr(), None) < 0


# ----------------------------------------------------------- small grids: deep LDS-DMA rings, K-split tile
SMALL_GRID_CASES = [
    # n, cin, cout, k, stride, pad, H, W
    (1, 384, 256, (1, 5), 1, (0, 2), 32, 32),      # GRU z|r at batch 1: K-split tile, 6-deep ring
    (1, 384, 128, (5, 1), 1, (2, 0), 32, 32),
    (1, 128, 512, (3, 3), 1, 1, 32, 32),           # heads at batch 1
    (1, 324, 256, (1, 1), 1, 0, 32, 32),           # dense 1x1, ragged last chunk (324 = 10*32 + 4)
    (1, 64, 64, (3, 3), 1, 1, 128, 128),           # encoder at batch 1: pixel-split tile, 4-deep ring
    (1, 96, 96, (3, 3), 1, 1, 64, 64),
    (1, 64, 96, (3, 3), 2, 1, 128, 128),           # stride 2
    (2, 224, 128, (3, 3), 2, 1, 32, 32),           # pose head conv 1
    (1, 128, 128, (3, 3), 2, 1, 8, 8),             # pose head conv 3: 4 blocks
    (1, 256, 192, (3, 3), 1, 1, 12, 20),           # ragged tiles
    (3, 40, 33, (3, 3), 1, 1, 9, 7),               # ragged everything
]


@pytest.mark.parametrize('case', SMALL_GRID_CASES)
def test_conv2d_small_grid_kernels(case):
    """batch-1-sized grids run on the LDS-DMA kernel's deep-ring / K-split variants: same numbers as
    torch within fp32 round-off, with bias + ReLU, a residual, and the two-segment GRU-Q form."""
    import ctypes as C
    from scflow_amd import _lib
    n, cin, cout, k, stride, pad, H, W = case
    x = rnd((n, cin, H, W), 80)
    w = rnd((cout, cin, *k), 81, 0.05)
    b = rnd((cout,), 82, 0.2)
    pc = ops.PackedConv.from_weight(w.to(DEV), b.to(DEV), stride=stride, padding=pad)
    want = torch.relu(F.conv2d(x, w, b, stride=stride, padding=pad))
    xd = x.to(DEV)
    got = ops.conv2d(pc, xd, act=ops.ACT_RELU)
    close(got, want, atol=3e-5 * max(1.0, float(want.abs().max())), what='bias+relu')
    # the library really took the LDS-DMA path (info[3] < 0 with the a4 packing present)
    d = _lib.ConvDesc()
    d.in0, d.C0, d.in0_nstride = xd.data_ptr(), cin, cin * H * W
    d.N, d.H, d.W = n, H, W
    d.wp, d.Mld, d.Cout, d.KC = pc.wp.data_ptr(), pc.mld, cout, pc.kc
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w = pc.kh, pc.kw, stride, pc.pad_h, pc.pad_w
    d.out, d.out_nstride, d.out_div = got.data_ptr(), got.stride(0), 1.0
    if pc.wp4 is not None:
        d.wp_a4, d.a4_groups, d.a4_mld = pc.wp4.data_ptr(), pc.g4, pc.mld
    if pc.wp4s is not None:
        d.wp_a4s, d.a4s_groups, d.a4_mld = pc.wp4s.data_ptr(), pc.g4s, pc.mld
    info = (C.c_int32 * 4)()
    assert _lib.load().scf_conv2d_query(C.byref(d), info) == 0 and info[3] < 0, list(info)
    # residual + no activation, output into a channel slice
    ho, wo = want.shape[-2:]
    res = rnd((n, cout, ho, wo), 83)
    big = torch.zeros((n, cout + 5, ho, wo), device=DEV)
    ops.conv2d(pc, xd, out=big[:, 3:3 + cout], res=res.to(DEV))
    close(big[:, 3:3 + cout], F.conv2d(x, w, b, stride=stride, padding=pad) + res,
          atol=3e-5 * max(1.0, float(want.abs().max())), what='residual into slice')
    assert float(big[:, :3].abs().max()) == 0 and float(big[:, 3 + cout:].abs().max()) == 0
    # same result as the register-staged kernel (no DMA packing) within round-off
    pc0 = ops.PackedConv.from_weight(w.to(DEV), b.to(DEV), stride=stride, padding=pad, dma_packing=False)
    close(ops.conv2d(pc0, xd, act=ops.ACT_RELU), want, atol=3e-5 * max(1.0, float(want.abs().max())), what='register-staged')


# ------------------------------------------------------------------ K split across blocks (r5)
@pytest.mark.parametrize('n,cin,cout,hw,ks', [(1, 224, 128, 32, 4), (3, 128, 128, 16, 4), (32, 128, 128, 8, 2),
                                              (2, 224, 128, 30, 3), (32, 224, 128, 32, 4), (5, 96, 64, 12, 2)])
def test_conv2d_kslices_partial_tensors(n, cin, cout, hw, ks):
    """scf_conv_desc.k_slices: S groups of blocks each contract 1 / S of the channel chunks into their own partial tensor.
    The partial tensors add up (in slice order) to the unsliced launch's result within the re-association error, and to
    torch fp64 within the direct kernels' tolerance; GroupNorm on the partial tensors is bit-identical to GroupNorm on
    their ordered sum (what the pose head relies on)."""
    x = rnd((n, cin, hw, hw), 31 + n).abs().to(DEV)
    wt = rnd((cout, cin, 3, 3), 32 + n, (1.0 / (cin * 9)) ** 0.5).to(DEV)
    pc = ops.PackedConv.from_weight(wt, None, stride=2, padding=1)
    want = F.conv2d(x.double(), wt.double(), None, stride=2, padding=1)
    with ops.record_conv_kernels() as ran:
        parts = ops.conv2d(pc, x, kslices=ks)
    assert [k for _, k in ran] == ['direct-dma'], ran
    assert parts.shape == (ks, n) + tuple(want.shape[1:])
    total = parts[0].clone()
    for s_ in range(1, ks):
        total += parts[s_]
    base = ops.conv2d(pc, x)
    scale = F.conv2d(x.double().abs(), wt.double().abs(), None, stride=2, padding=1)
    eps = 2.0 ** -24
    r = float(((total.double() - want).abs() / (eps * scale)).max())
    rb = float(((base.double() - want).abs() / (eps * scale)).max())
    print(f'[measured] kslices={ks} N{n} {cin}->{cout} @{hw // 2}: err / (eps sum|w||x|) = {r:.2f} (unsliced {rb:.2f})')
    assert r <= 24.0 and rb <= 24.0
    g, b = (rnd((cout,), 5).abs() + 0.5).to(DEV), rnd((cout,), 6).to(DEV)
    a = ops.group_norm_relu(parts, g, b, 32)
    ref = ops.group_norm_relu(total, g, b, 32)
    assert torch.equal(a, ref)
    # every element of every partial tensor was written (no stale memory where a slice owns few chunks)
    parts2 = ops.conv2d(pc, x, kslices=ks, out=torch.full_like(parts, float('nan')))
    assert torch.equal(parts, parts2)


def test_conv2d_kslices_rejects_epilogues():
    x = rnd((1, 128, 8, 8), 1).to(DEV)
    wt = rnd((128, 128, 3, 3), 2, 0.03).to(DEV)
    with pytest.raises(ops._lib.ScflowHipError):
        ops.conv2d(ops.PackedConv.from_weight(wt, rnd((128,), 3).to(DEV), stride=2, padding=1), x, kslices=2)
    with pytest.raises(ops._lib.ScflowHipError):
        ops.conv2d(ops.PackedConv.from_weight(wt, None, stride=2, padding=1), x, kslices=2, act=ops.ACT_RELU)


def test_pose_head_kslices_policy_and_parity(golden_dir):
    """the pose head with its convolutions K-sliced (ops.conv_kslices) against the unsliced head: same features within
    1e-5 relative, at batch 1 (4 slices per layer) and batch 32 (4 / 2 / 2)."""
    import json, os, scflow_amd
    shapes = json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))['shapes']
    m = scflow_amd.build_refiner(scflow_amd.scflow_model_cfg())
    m.load_state_dict(scflow_amd.fill_state_dict(shapes, seed=3), strict=True)
    ph = m.to(DEV).decoder.pose_pred
    for n in (1, 32):
        x0, x1 = rnd((n, 128, 32, 32), 9).to(DEV), rnd((n, 96, 32, 32), 10).abs().to(DEV)
        ks = [ops.conv_kslices(b.packed, n, hh, hh) for b, hh in zip(ph.conv_layers, (32, 16, 8))]
        assert ks == ([4, 4, 4] if n == 1 else [4, 2, 2]), ks
        ra, ta = ph.features(x0, x1)
        prev = ops.set_conv_kslices(False)
        try:
            rb, tb = ph.features(x0, x1)
        finally:
            ops.set_conv_kslices(prev)
        for a, b in ((ra, rb), (ta, tb)):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))


def test_corr_lookup_groups_per_block_are_bit_identical():
    """r5: at >= 4 x CUs groups the dispatch packs four groups of 32 queries into one 1024-thread block (the same waves in a
    quarter of the workgroups); every packing behind the knob (one / two / four groups per block, the pipelined variants)
    returns the same bits as the one-group kernel, on both pyramid layouts, incl. a ragged last group."""
    for (n, h, w) in ((32, 32, 32), (33, 32, 32), (5, 24, 40)):
        f1, f2 = rnd((n, 64, h, w), 70).to(DEV), rnd((n, 64, h, w), 71).to(DEV)
        flow = (rnd((n, 2, h, w), 72) * 4).to(DEV)
        flow[0, :, 0, 0] = 1e6
        for mask in (0, ops.pyramid_layout(h, w, 4, 4)):
            pyr = ops.corr_build(f1, f2, 4, tiled_levels=mask)
            try:
                ops.tune('lookup_pipe', 1)
                want = ops.corr_lookup(pyr, flow, 4, tiled_levels=mask)
                for mode in (0, 2, 3, 4, 5, 6):
                    ops.tune('lookup_pipe', mode)
                    got = ops.corr_lookup(pyr, flow, 4, tiled_levels=mask)
                    assert torch.equal(got, want), (n, h, w, mask, mode)
            finally:
                ops.tune('lookup_pipe', 0)


def test_conv2d_autoslice_small_grids():
    """r5: with a workspace registered on the launch stream (ops.register_conv_workspace: opt-in) the library splits small-grid
    convolutions into K slices + a combine launch that runs the layer's own epilogue.  Same result as the single launch up to
    the re-association of <= 4 partial sums -- for the affine, sigmoid / tanh and both GRU epilogues, one- and two-segment
    inputs -- and run-to-run identical; scf_tune('conv_autoslice', 0) is the single-launch path."""
    n, H, W = 1, 32, 32
    eps = 2.0 ** -24

    def both(fn):
        ops.register_conv_workspace(True)        # opt-in: nothing registers a workspace by default (measured: a loss here)
        try:
            a = fn()
            a2 = fn()
            prev = ops.tune('conv_autoslice', 0)
            try:
                b = fn()
            finally:
                ops.tune('conv_autoslice', prev)
        finally:
            ops.register_conv_workspace(False)
        c = fn()                                 # cleared: the single launch again
        assert torch.equal(b, c)
        return a, a2, b

    # (cin, cout, k, pad, act): the batch-1 layers whose chains are long enough to be sliced
    for cin, cout, k, pad, act in ((256, 192, (3, 3), 1, ops.ACT_RELU), (256, 126, (3, 3), 1, ops.ACT_RELU),
                                   (324, 256, (1, 1), 0, ops.ACT_RELU), (256, 64, (3, 3), 1, ops.ACT_TANH)):
        x = rnd((n, cin, H, W), 80 + cout).abs().to(DEV)
        wt = rnd((cout, cin) + k, 81 + cout, (1.0 / (cin * k[0] * k[1])) ** 0.5).to(DEV)
        b = rnd((cout,), 82, 0.1).to(DEV)
        pc = ops.PackedConv.from_weight(wt, b, padding=pad)
        got, again, single = both(lambda: ops.conv2d(pc, x, act=act))
        assert torch.equal(got, again)
        assert not torch.equal(got, single), 'the launch was not sliced (rule or workspace missing)'
        want = F.conv2d(x.double(), wt.double(), b.double(), padding=pad)
        want = torch.relu(want) if act == ops.ACT_RELU else torch.tanh(want)
        scale = F.conv2d(x.double().abs(), wt.double().abs(), b.double().abs(), padding=pad)
        for t, what in ((got, 'sliced'), (single, 'single')):
            r = float(((t.double() - want).abs() / (eps * scale)).max())
            assert r <= 24.0, (what, cin, cout, r)
    # the whole GRU cell (z|r and q epilogues, hoisted context term) at batch 1
    from scflow_amd.modules import ConvGRU
    torch.manual_seed(5)
    gru = ConvGRU(128, 256, 'SeqConv').to(DEV)
    hx0 = rnd((n, 128 + 128 + 128, H, W), 90).to(DEV)
    hx0[:, :128] = torch.tanh(hx0[:, :128])
    hx0[:, 128:] = hx0[:, 128:].abs()

    def cell():
        hx = hx0.clone()
        ctx = gru.context_terms(hx[:, 128:256])
        for _ in range(3):
            gru.forward_inplace(hx, ctx, 128)
        return hx[:, :128].clone()
    got, again, single = both(cell)
    assert torch.equal(got, again)
    d = float((got - single).abs().max())
    print(f'[measured] GRU cell x3 at batch 1, K-sliced vs single launches: max |dh| {d:.2e}')
    assert 0.0 < d <= 2e-5


# ------------------------------------------------------------------------------------------------
# r6: two independent small-grid layers in one launch (scf_conv2d_pair)
# ------------------------------------------------------------------------------------------------
PAIR_CASES = [
    # (n, cin, cout, k, pad, hw) x 2, expect merged?
    ((1, 128, 64, (3, 3), (1, 1), (32, 32)), (1, 64, 32, (3, 3), (1, 1), (32, 32)), True),      # delta-flow | mask encoders, 2nd layers
    ((1, 2, 128, (7, 7), (3, 3), (32, 32)), (1, 1, 64, (3, 3), (1, 1), (32, 32)), True),        # ... 1st layers (thin inputs)
    ((1, 256, 192, (3, 3), (1, 1), (32, 32)), (1, 128, 64, (3, 3), (1, 1), (32, 32)), True),    # corr_net.1 | flow_net.1 at batch 1
    ((1, 256, 2, (3, 3), (1, 1), (32, 32)), (1, 256, 1, (1, 1), (0, 0), (32, 32)), True),       # flow | mask predictions (two thin instantiations)
    ((2, 256, 192, (3, 3), (1, 1), (32, 32)), (2, 128, 64, (3, 3), (1, 1), (32, 32)), False),   # batch 2: 384 + 128 blocks > CUs
    ((1, 324, 256, (1, 1), (0, 0), (32, 32)), (1, 2, 128, (7, 7), (3, 3), (32, 32)), True),     # corr_net.0 | flow_net.0: K-split tile + thin-input kernel
    ((1, 2, 128, (7, 7), (3, 3), (32, 32)), (1, 128, 64, (3, 3), (1, 1), (32, 32)), True),      # the same two families, thin-input layer first
    ((1, 128, 64, (3, 3), (1, 1), (32, 32)), (1, 256, 2, (3, 3), (1, 1), (32, 32)), False),     # families without a shared launch (K-split | thin-output)
    ((32, 128, 64, (3, 3), (1, 1), (32, 32)), (32, 64, 32, (3, 3), (1, 1), (32, 32)), False),   # full grids (Winograd)
    ((3, 224, 128, (3, 3), (1, 1), (12, 20)), (2, 30, 40, (1, 5), (0, 2), (9, 33)), None),      # ragged shapes, whatever it does
]


@pytest.mark.parametrize('case', PAIR_CASES)
def test_conv2d_pair_equals_two_launches(case):
    """scf_conv2d_pair == scf_conv2d(a); scf_conv2d(b) bit for bit, merged or not; the dispatch log names both layers."""
    (na, cia, coa, ka, pa, hwa), (nb, cib, cob, kb, pb, hwb), merged = case
    xa, xb = rnd((na, cia, *hwa), 201).to(DEV), rnd((nb, cib, *hwb), 202).to(DEV)
    wa = rnd((coa, cia, *ka), 203, (1.0 / (cia * ka[0] * ka[1])) ** 0.5).to(DEV)
    wb = rnd((cob, cib, *kb), 204, (1.0 / (cib * kb[0] * kb[1])) ** 0.5).to(DEV)
    pca = ops.PackedConv.from_weight(wa, rnd((coa,), 205, 0.1).to(DEV), padding=pa)
    pcb = ops.PackedConv.from_weight(wb, rnd((cob,), 206, 0.1).to(DEV), padding=pb)
    want_a = ops.conv2d(pca, xa, act=ops.ACT_RELU)
    want_b = ops.conv2d(pcb, xb, act=ops.ACT_NONE)
    with ops.record_conv_kernels() as ran:
        got_a, got_b = ops.conv2d_pair((pca, xa, dict(act=ops.ACT_RELU)), (pcb, xb, dict(act=ops.ACT_NONE)))
    torch.cuda.synchronize()
    assert torch.equal(got_a, want_a) and torch.equal(got_b, want_b)
    assert len(ran) == 2 and ran[0][0].startswith(f'{cia}->{coa}') and ran[1][0].startswith(f'{cib}->{cob}'), ran
    close(got_a, torch.relu(F.conv2d(xa.cpu(), wa.cpu(), pca.bias.cpu(), padding=pa)), atol=3e-5, what='pair a vs torch')
    close(got_b, F.conv2d(xb.cpu(), wb.cpu(), pcb.bias.cpu(), padding=pb), atol=3e-5, what='pair b vs torch')
