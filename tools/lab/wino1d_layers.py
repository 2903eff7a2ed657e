"""The SepConvGRU's 1x5 / 5x1 layer shapes at batch 32 (and on 60x80 maps): direct kernel vs the F(2, 5)
Winograd kernel, launch-bound timers, plain epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops
DEV = 'cuda:0'
cases = [('zr 256->256 1x5 N32', 32, 256, 256, (1, 5), 32, 32), ('zr 256->256 5x1 N32', 32, 256, 256, (5, 1), 32, 32),
         ('q 256->128 1x5 N32', 32, 256, 128, (1, 5), 32, 32), ('q 256->128 5x1 N32', 32, 256, 128, (5, 1), 32, 32),
         ('zr 256->256 1x5 N8 60x80', 8, 256, 256, (1, 5), 60, 80), ('q 256->128 5x1 N8 60x80', 8, 256, 128, (5, 1), 60, 80),
         ('zr 256->256 1x5 N8', 8, 256, 256, (1, 5), 32, 32), ('q 256->128 5x1 N8', 8, 256, 128, (5, 1), 32, 32)]
for name, n, cin, cout, k, H, W in cases:
    pad = (0, 2) if k == (1, 5) else (2, 0)
    x = torch.randn((n, cin, H, W), device=DEV)
    w = torch.randn((cout, cin, *k), device=DEV) * (1.0 / (cin * 5)) ** 0.5
    b = torch.randn((cout,), device=DEV) * 0.1
    pc = ops.PackedConv.from_weight(w, b, padding=pad)
    out = torch.empty((n, cout, H, W), device=DEV)
    fl = 2.0 * n * cout * cin * 5 * H * W
    res = []
    for wino in (False, True):
        prev = ops.set_conv_winograd(wino)
        for _ in range(50):
            ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)
        ts = sorted(ops.time_first_kernel(lambda: ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)) for _ in range(7))
        ops.set_conv_winograd(prev)
        res.append(ts[3])
    print(f'{name:28s} direct {res[0]:8.1f} us {fl / res[0] * 1e-6:6.1f} TF/s | F(2,5) {res[1]:8.1f} us '
          f'{fl / res[1] * 1e-6:6.1f} TF/s (direct-equivalent)  x{res[0] / res[1]:.2f}', flush=True)
