"""The SepConvGRU's 1x5 / 5x1 layer shapes: direct kernel vs F(2, 5) vs F(4, 5) (conv_wino1d4.hip) -- launch-bound
timers and the max error vs torch fp64 (CPU) in units of eps * sum|w||x|; then the whole cell (gate epilogues)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops
DEV = 'cuda:0'
cases = [('zr 384->256 1x5 N32', 32, 384, 256, (1, 5), 32, 32), ('zr 384->256 5x1 N32', 32, 384, 256, (5, 1), 32, 32),
         ('q 384->128 1x5 N32', 32, 384, 128, (1, 5), 32, 32), ('q 384->128 5x1 N32', 32, 384, 128, (5, 1), 32, 32),
         ('zr 256->256 1x5 N8 60x80', 8, 256, 256, (1, 5), 60, 80), ('zr 256->256 5x1 N8 60x80', 8, 256, 256, (5, 1), 60, 80),
         ('ragged 40->64 1x5 N32 20x30', 32, 40, 64, (1, 5), 20, 30), ('ragged 48->64 5x1 N24 21x28', 24, 48, 64, (5, 1), 21, 28)]
torch.manual_seed(0)
for name, n, cin, cout, k, H, W in cases:
    pad = (0, 2) if k == (1, 5) else (2, 0)
    x = torch.randn((n, cin, H, W))
    w = torch.randn((cout, cin, *k)) * (1.0 / (cin * 5)) ** 0.5
    b = torch.randn((cout,)) * 0.1
    ne = min(n, 2)
    want = torch.relu(F.conv2d(x[:ne].double(), w.double(), b.double(), padding=pad))
    scale = F.conv2d(x[:ne].double().abs(), w.double().abs(), b.double().abs(), padding=pad) * 2.0 ** -24
    pc = ops.PackedConv.from_weight(w.to(DEV), b.to(DEV), padding=pad)
    xd = x.to(DEV)
    out = torch.empty((n, cout, H, W), device=DEV)
    fl = 2.0 * n * cout * cin * 5 * H * W
    res = []
    for wino, w4 in ((False, 0), (True, 0), (True, 2)):
        prev = ops.set_conv_winograd(wino)
        ops.tune('wino1d4', w4)
        with ops.record_conv_kernels() as ran:
            ops.conv2d(pc, xd, out=out, act=ops.ACT_RELU)
        err = float(((out[:ne].cpu().double() - want).abs() / scale).max())
        for _ in range(30):
            ops.conv2d(pc, xd, out=out, act=ops.ACT_RELU)
        ts = sorted(ops.time_first_kernel(lambda: ops.conv2d(pc, xd, out=out, act=ops.ACT_RELU)) for _ in range(7))
        ops.set_conv_winograd(prev)
        ops.tune('wino1d4', 1)
        res.append((ts[3], err, ran[0][1]))
    print(f'{name:30s} ' + ' | '.join(f'{kk:16s} {t:7.1f} us {fl / t * 1e-6:6.1f} TF/s err {e:5.1f}' for t, e, kk in res)
          + f'  F(4,5)/F(2,5) x{res[1][0] / res[2][0]:.2f}', flush=True)

# the whole cell at batch 32: both gate epilogues + the context term through the accumulators
from scflow_amd.modules import ConvGRU
torch.manual_seed(12)
n, h, w = 32, 32, 32
hc, cc, xc = 128, 128, 128
gru = ConvGRU(hc, cc + xc, 'SeqConv').to(DEV)
hx = torch.randn((n, hc + cc + xc, h, w))
hx[:, :hc] = torch.tanh(hx[:, :hc])
hx[:, hc:] = torch.relu(hx[:, hc:] + 0.5)
hist = {}
for tag, wino, w4 in (('direct', False, 0), ('F(2,5)', True, 0), ('F(4,5)', True, 1)):
    prev = ops.set_conv_winograd(wino)
    ops.tune('wino1d4', w4)
    gru.invalidate_packed()
    a = hx.to(DEV)
    ctx = gru.context_terms(a[:, hc:hc + cc])
    states = []
    with ops.record_conv_kernels() as ran:
        for it in range(6):
            gru.forward_inplace(a, ctx, cc)
            states.append(a[:, :hc].clone())
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for it in range(50):
        gru.forward_inplace(a, ctx, cc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50 * 1e3
    hist[tag] = states
    print(tag, sorted(set(k for _, k in ran)), f'{dt:.3f} ms per cell update (4 launches)')
    ops.set_conv_winograd(prev)
    ops.tune('wino1d4', 1)
for tag in ('F(2,5)', 'F(4,5)'):
    print(tag, 'vs direct, max |dh| per iteration:', ' '.join(f'{float((a_ - b_).abs().max()):.1e}' for a_, b_ in zip(hist[tag], hist['direct'])))
