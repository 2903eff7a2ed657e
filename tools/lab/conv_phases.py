"""Phase ablation of the direct LDS-DMA convolution kernel (tools/lab/build_conv_masks.sh): kernel time of the
GRU-sized and the large 3x3 layers with the in-loop copies / the MFMAs / the epilogue compiled out (results
are then wrong; only the durations mean something).  SCF_EXP_SUFFIX=_c<mask> selects the build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', f'libscflow_hip_exp{os.environ.get("SCF_EXP_SUFFIX", "")}.so')
from scflow_amd import ops
DEV = 'cuda:0'
ops.set_conv_winograd(False)
cases = [('GRU zr 256->256 1x5', 32, 256, 256, (1, 5), (0, 2), 32), ('GRU q 256->128 5x1', 32, 256, 128, (5, 1), (2, 0), 32),
         ('128->512 3x3', 32, 128, 512, (3, 3), 1, 32), ('1x1 324->256', 32, 324, 256, (1, 1), 0, 32)]
out = []
for name, n, cin, cout, k, pad, hw in cases:
    x = torch.randn((n, cin, hw, hw), device=DEV)
    w = torch.randn((cout, cin, *k), device=DEV) * (1.0 / (cin * k[0] * k[1])) ** 0.5
    b = torch.randn((cout,), device=DEV) * 0.1
    pc = ops.PackedConv.from_weight(w, b, padding=pad)
    o = torch.empty((n, cout, hw, hw), device=DEV)
    for _ in range(200):
        ops.conv2d(pc, x, out=o, act=ops.ACT_RELU)
    ts = sorted(ops.time_first_kernel(lambda: ops.conv2d(pc, x, out=o, act=ops.ACT_RELU)) for _ in range(7))
    out.append(f'{name}: {ts[3]:7.1f} us')
print(os.environ.get('SCF_EXP_SUFFIX', ''), ' | '.join(out))
