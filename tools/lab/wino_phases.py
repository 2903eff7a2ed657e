"""Phase ablation of the Winograd kernel (library built with -DSCF_WINO_LAB, tools/lab/build_exp.sh): the kernel
time with the MFMAs / the input transform / the in-loop DMA / the stores / the per-chunk barrier removed
(results are then wrong; only the durations mean something)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', f'libscflow_hip_exp{os.environ.get("SCF_EXP_SUFFIX", "")}.so')
from scflow_amd import ops
DEV = 'cuda:0'
cases = [('128->512 @32 N32', 32, 128, 512, 32, 32), ('64->64 @128 N64', 64, 64, 64, 128, 128), ('96->96 @64 N64', 64, 96, 96, 64, 64)]
if os.environ.get('SCF_EXP_SUFFIX'):
    masks = [(0, 'compile-time mask ' + os.environ['SCF_EXP_SUFFIX'])]
else:
  masks = [(0, 'full'), (1, 'no MFMA'), (2, 'no transform'), (4, 'no loop DMA'), (8, 'no stores'), (16, 'no barrier'),
         (3, 'no MFMA, no transform'), (6, 'no transform, no DMA'), (7, 'no MFMA/transform/DMA'), (1 + 2 + 4 + 8, 'loop skeleton only')]
ops.set_conv_winograd(True)
for name, n, cin, cout, H, W in cases:
    x = torch.randn((n, cin, H, W), device=DEV)
    w = torch.randn((cout, cin, 3, 3), device=DEV) * (1.0 / (cin * 9)) ** 0.5
    b = torch.randn((cout,), device=DEV) * 0.1
    pc = ops.PackedConv.from_weight(w, b, padding=1)
    out = torch.empty((n, cout, H, W), device=DEV)
    print(name)
    for m, what in masks:
        os.environ['SCF_WINO_LAB'] = str(m)
        for _ in range(300):              # ~60 ms: the shader clock takes a while to ramp up
            ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)
        ts = sorted(ops.time_first_kernel(lambda: ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)) for _ in range(5))
        print(f'   {what:28s} {ts[2]:8.1f} us', flush=True)
