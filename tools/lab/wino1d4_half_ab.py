"""F(4, 5): the half-domain kernel (scf_tune wino1d4_half = 1: a wave holds 4 of the 8 transform positions for two channel
fragments) against the full-domain kernel of r4 (= 0): error vs fp64 in units of eps * sum|w||x|, max |difference| between the
two, launch-bound times per GRU layer shape, then the whole cell (gate epilogues + context term) and the batch-32 step."""
import os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops
DEV = 'cuda:0'
cases = [('zr 384->256 1x5 N32', 32, 384, 256, (1, 5), 32, 32), ('zr 384->256 5x1 N32', 32, 384, 256, (5, 1), 32, 32),
         ('zr 256->256 1x5 N32', 32, 256, 256, (1, 5), 32, 32), ('zr 256->256 5x1 N32', 32, 256, 256, (5, 1), 32, 32),
         ('q 256->128 1x5 N32', 32, 256, 128, (1, 5), 32, 32), ('q 256->128 5x1 N32', 32, 256, 128, (5, 1), 32, 32),
         ('zr 256->256 1x5 N8 60x80', 8, 256, 256, (1, 5), 60, 80), ('zr 256->256 5x1 N8 60x80', 8, 256, 256, (5, 1), 60, 80),
         ('ragged 40->64 1x5 N32 20x30', 32, 40, 64, (1, 5), 20, 30), ('ragged 48->64 5x1 N24 21x28', 24, 48, 64, (5, 1), 21, 28),
         ('ragged 36->128 1x5 N16 23x27', 16, 36, 128, (1, 5), 23, 27)]
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
    res, outs = [], []
    ops.tune('wino1d4', 2)
    for hv in (0, 1):
        ops.tune('wino1d4_half', hv)
        out = torch.empty((n, cout, H, W), device=DEV)
        with ops.record_conv_kernels() as ran:
            ops.conv2d(pc, xd, out=out, act=ops.ACT_RELU)
        err = float(((out[:ne].cpu().double() - want).abs() / scale).max())
        for _ in range(30):
            ops.conv2d(pc, xd, out=out, act=ops.ACT_RELU)
        ts = sorted(ops.time_first_kernel(lambda: ops.conv2d(pc, xd, out=out, act=ops.ACT_RELU)) for _ in range(9))
        res.append((ts[4], err, ran[0][1]))
        outs.append(out)
    ops.tune('wino1d4', 1); ops.tune('wino1d4_half', 1)
    diff = float((outs[0] - outs[1]).abs().max())
    fl = 2.0 * n * cout * cin * 5 * H * W * 0.4
    print(f'{name:30s} ' + ' | '.join(f'{("full", "half")[i]} {t:7.1f} us {fl / t * 1e-6:6.1f} TF/s exec err {e:5.1f} ({kk})' for i, (t, e, kk) in enumerate(res))
          + f'  half/full x{res[0][0] / res[1][0]:.3f}  max|full - half| {diff:.1e}', flush=True)

from scflow_amd.modules import ConvGRU
torch.manual_seed(12)
n, h, w = 32, 32, 32
hc, cc, xc = 128, 128, 128
gru = ConvGRU(hc, cc + xc, 'SeqConv').to(DEV)
hx = torch.randn((n, hc + cc + xc, h, w))
hx[:, :hc] = torch.tanh(hx[:, :hc])
hx[:, hc:] = torch.relu(hx[:, hc:] + 0.5)
hist = {}
for rep in range(2):
    for tag, wino, hv in (('direct', False, 0), ('F(4,5) full', True, 0), ('F(4,5) half', True, 1)):
        prev = ops.set_conv_winograd(wino)
        ops.tune('wino1d4_half', hv)
        gru.invalidate_packed()
        a = hx.to(DEV)
        ctx = gru.context_terms(a[:, hc:hc + cc])
        states = []
        with ops.record_conv_kernels() as ran:
            for it in range(6):
                gru.forward_inplace(a, ctx, cc)
                states.append(a[:, :hc].clone())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(100):
            gru.forward_inplace(a, ctx, cc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100 * 1e3
        hist[tag] = states
        print(f'rep {rep}', tag, sorted(set(k for _, k in ran)), f'{dt:.4f} ms per cell update (4 launches)', flush=True)
        ops.set_conv_winograd(prev)
        ops.tune('wino1d4_half', 1)
for tag in ('F(4,5) full', 'F(4,5) half'):
    print(tag, 'vs direct, max |dh| per iteration:', ' '.join(f'{float((a_ - b_).abs().max()):.1e}' for a_, b_ in zip(hist[tag], hist['direct'])))
