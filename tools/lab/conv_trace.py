"""Per-chunk timeline of the small-grid (K-split) convolution kernel: block 0's four waves stamp
s_memrealtime (10 ns) at entry, after setup, after the prologue staging, and per chunk after the
vmcnt wait / the barrier / the staging of chunk c+NST-1 / the MFMA phase.  Needs the library built
with -DSCF_CONV_LAB (tools/lab/build_exp.sh)."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops, _lib
DEV = 'cuda:0'
lib = _lib.load()
lib.scf_conv_trace_set.argtypes = [C.c_void_p, C.c_int]
cases = [('GRU zr 384->256 1x5', 1, 384, 256, (1, 5), 1, (0, 2), 32, 32),
         ('heads 128->512 3x3', 1, 128, 512, (3, 3), 1, 1, 32, 32),
         ('pose conv3 128->128 3x3/s2 @4x4', 1, 128, 128, (3, 3), 2, 1, 8, 8),
         ('corr 324->256 1x1', 1, 324, 256, (1, 1), 1, 0, 32, 32)]
for name, n, cin, cout, k, stride, pad, H, W in cases:
    x = torch.randn((n, cin, H, W), device=DEV)
    w = torch.randn((cout, cin, *k), device=DEV) * 0.05
    b = torch.randn((cout,), device=DEV)
    pc = ops.PackedConv.from_weight(w, b, stride=stride, padding=pad)
    for _ in range(3):
        ops.conv2d(pc, x, act=ops.ACT_RELU)
    tr = torch.zeros((4, 128), dtype=torch.int64, device=DEV)
    flush = torch.zeros(64 << 20, device=DEV)
    flush.add_(1.0)                                   # weights out of L2
    torch.cuda.synchronize()
    lib.scf_conv_trace_set(C.c_void_p(tr.data_ptr()), 1)
    us = ops.time_first_kernel(lambda: ops.conv2d(pc, x, act=ops.ACT_RELU))
    torch.cuda.synchronize()
    lib.scf_conv_trace_set(None, 0)
    t = tr.cpu()
    t0 = int(t[:, 0].min())
    rel = lambda v: (int(v) - t0) * 0.01
    print(f'== {name}: kernel {us:.1f} us')
    for wv in range(4):
        row = t[wv]
        nch = sum(1 for c in range(30) if int(row[4 + 4 * c]))
        s = f'  wave {wv}: entry {rel(row[0]):.2f} setup {rel(row[1]):.2f} prologue {rel(row[2]):.2f} |'
        for c in range(nch):
            a, bb, cc, d = (rel(row[4 + 4 * c + i]) for i in range(4))
            s += f' c{c}: wait {a:.2f} bar {bb:.2f} stage {cc:.2f} mfma {d:.2f} |'
        s += f' end {rel(row[3]):.2f}'
        print(s[:1500])
