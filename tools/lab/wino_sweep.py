"""Where the Winograd kernel starts to win: every 3x3 stride-1 layer shape of the refiner at batch 1 ... 32,
direct vs Winograd kernel time, with the Winograd grid size (blocks)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops, _lib
DEV = 'cuda:0'
lib = _lib.load()
shapes = [('64->64 @128', 2, 64, 64, 128), ('96->96 @64', 2, 96, 96, 64), ('128->128 @32', 2, 128, 128, 32),
          ('128->512 @32', 1, 128, 512, 32), ('256->192 @32', 1, 256, 192, 32), ('256->126 @32', 1, 256, 126, 32),
          ('128->64 @32', 1, 128, 64, 32)]
for name, mult, cin, cout, hw in shapes:
    w = torch.randn((cout, cin, 3, 3), device=DEV) * (1.0 / (cin * 9)) ** 0.5
    b = torch.randn((cout,), device=DEV) * 0.1
    pc = ops.PackedConv.from_weight(w, b, padding=1)
    for batch in (1, 2, 4, 8, 16, 32):
        n = batch * mult
        x = torch.randn((n, cin, hw, hw), device=DEV)
        out = torch.empty((n, cout, hw, hw), device=DEV)
        res = []
        for wino in (False, True):
            ops.set_conv_winograd(wino)
            for _ in range(20):
                ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)
            ts = sorted(ops.time_first_kernel(lambda: ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)) for _ in range(9))
            res.append(ts[4])
        ops.set_conv_winograd(True)
        d, _ = ops.conv_desc(pc, x, out=out, act=ops.ACT_RELU)
        info = (C.c_int32 * 4)()
        lib.scf_conv2d_query(C.byref(d), info)
        print(f'{name:14s} batch {batch:2d} (N {n:2d}) blocks {info[2]:5d}: direct {res[0]:7.1f} us, winograd {res[1]:7.1f} us  x{res[0] / res[1]:.2f}', flush=True)
