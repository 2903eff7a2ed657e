"""Per-chunk timeline of the full-grid (pixel-split) convolution tiles at batch 32: block 0's four
waves (see conv_trace.py).  Needs the library built with -DSCF_CONV_LAB."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops, _lib
DEV = 'cuda:0'
lib = _lib.load()
lib.scf_conv_trace_set.argtypes = [C.c_void_p, C.c_int]
cases = [('heads 128->512 3x3', 32, 128, 512, (3, 3), 1, 1, 32, 32),
         ('GRU zr 384->256 5x1', 32, 384, 256, (5, 1), 1, (2, 0), 32, 32),
         ('enc 64->64 3x3 @128', 64, 64, 64, (3, 3), 1, 1, 128, 128),
         ('flow 128->64 3x3', 32, 128, 64, (3, 3), 1, 1, 32, 32)]
for name, n, cin, cout, k, stride, pad, H, W in cases:
    x = torch.randn((n, cin, H, W), device=DEV)
    w = torch.randn((cout, cin, *k), device=DEV) * 0.05
    b = torch.randn((cout,), device=DEV)
    pc = ops.PackedConv.from_weight(w, b, stride=stride, padding=pad)
    for _ in range(3):
        ops.conv2d(pc, x, act=ops.ACT_RELU)
    tr = torch.zeros((4, 128), dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()
    lib.scf_conv_trace_set(C.c_void_p(tr.data_ptr()), 1)
    us = ops.time_first_kernel(lambda: ops.conv2d(pc, x, act=ops.ACT_RELU))
    torch.cuda.synchronize()
    lib.scf_conv_trace_set(None, 0)
    t = tr.cpu()
    t0 = int(t[:, 0].min())
    rel = lambda v: (int(v) - t0) * 0.01
    fl = 2.0 * n * cout * cin * k[0] * k[1] * (H // stride) * (W // stride)
    print(f'== {name}: kernel {us:.1f} us  {fl / us * 1e-6:.1f} TFLOP/s')
    for wv in range(4):
        row = t[wv]
        nch = sum(1 for c in range(30) if int(row[4 + 4 * c]))
        s = f'  wave {wv}: entry {rel(row[0]):.2f} setup {rel(row[1]):.2f} prologue {rel(row[2]):.2f} |'
        prev = rel(row[2])
        for c in range(nch):
            a, bb, cc, d = (rel(row[4 + 4 * c + i]) for i in range(4))
            s += f' c{c}: wait +{a - prev:.2f} bar +{bb - a:.2f} stage +{cc - bb:.2f} mfma +{d - cc:.2f} |'
            prev = d
        s += f' end {rel(row[3]):.2f} (epilogue +{rel(row[3]) - prev:.2f})'
        print(s[:2500])
