"""Every 3x3 / stride-1 layer shape of a batch-32 step (and of configs[4]): direct kernel vs the Winograd
F(2x2, 3x3) kernel, launch-bound timers (kernel durations)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops
DEV = 'cuda:0'
cases = [('64->64 @128 N64', 64, 64, 64, 128, 128), ('64->64 @128 N32', 32, 64, 64, 128, 128),
         ('96->96 @64 N64', 64, 96, 96, 64, 64), ('128->128 @32 N64', 64, 128, 128, 32, 32),
         ('128->512 @32 N32', 32, 128, 512, 32, 32), ('256->192 @32 N32', 32, 256, 192, 32, 32),
         ('256->126 @32 N32', 32, 256, 126, 32, 32), ('128->64 @32 N32', 32, 128, 64, 32, 32),
         ('128->512 @60x80 N8', 8, 128, 512, 60, 80), ('256->192 @60x80 N8', 8, 256, 192, 60, 80),
         ('128->512 @32 N1', 1, 128, 512, 32, 32), ('256->192 @32 N4', 4, 256, 192, 32, 32)]
if len(sys.argv) > 1:
    cases = [c for c in cases if any(a in c[0] for a in sys.argv[1:])]
for name, n, cin, cout, H, W in cases:
    x = torch.randn((n, cin, H, W), device=DEV)
    w = torch.randn((cout, cin, 3, 3), device=DEV) * (1.0 / (cin * 9)) ** 0.5
    b = torch.randn((cout,), device=DEV) * 0.1
    pc = ops.PackedConv.from_weight(w, b, padding=1)
    out = torch.empty((n, cout, H, W), device=DEV)
    fl = 2.0 * n * cout * cin * 9 * H * W
    res = []
    for wino in (False, True):
        prev = ops.set_conv_winograd(wino)
        for _ in range(3):
            ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)
        ts = sorted(ops.time_first_kernel(lambda: ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)) for _ in range(7))
        ops.set_conv_winograd(prev)
        res.append(ts[3])
    print(f'{name:24s} direct {res[0]:8.1f} us {fl / res[0] * 1e-6:6.1f} TF/s | winograd {res[1]:8.1f} us '
          f'{fl / res[1] * 1e-6:6.1f} TF/s (direct-equivalent)  x{res[0] / res[1]:.2f}', flush=True)
