"""EXPERIMENT (lever 1): the LDS-DMA convolution with channel-interleaved
activations [N][C/4][H][W][4] on its input and output, against the NCHW kernel on the same
layer.  Checks the result (after converting back) and reports both rates."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scflow_amd import _lib, ops
dev = 'cuda:0'


def to_c4(x):
    n, c, h, w = x.shape
    return x.view(n, c // 4, 4, h, w).permute(0, 1, 3, 4, 2).contiguous()


def from_c4(y, c):
    n = y.shape[0]
    return y.permute(0, 1, 4, 2, 3).reshape(n, c, y.shape[2], y.shape[3]).contiguous()


def run_c4(pc, wp4_c4, x4, out4, n, cin, h, w, act):
    d = _lib.ConvDesc()
    d.in0, d.C0, d.C1, d.in0_nstride = x4.data_ptr(), cin, 0, cin * h * w
    d.N, d.H, d.W = n, h, w
    d.wp, d.w_nstride, d.Mld, d.Cout = pc.wp.data_ptr(), 0, pc.mld, pc.cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.KC = pc.kh, pc.kw, pc.stride, pc.pad_h, pc.pad_w, pc.kc
    d.out, d.out_nstride = out4.data_ptr(), pc.cout * h * w
    d.bias = pc.bias.data_ptr()
    d.out_div, d.act = 1.0, act
    d.wp_a4, d.a4_groups, d.a4_mld = wp4_c4.data_ptr(), pc.g4, pc.mld
    d.in_c4, d.out_c4 = 1, 1
    _lib.check(_lib.load().scf_conv2d(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'conv c4')


def timeit(fn, n=15):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


for name, cin, cout, k, pad, H, W, n in (('heads 3x3 128>512', 128, 512, (3, 3), (1, 1), 32, 32, 32),
                                         ('gru 1x5 384>256', 384, 256, (1, 5), (0, 2), 32, 32, 32),
                                         ('gru 5x1 384>128', 384, 128, (5, 1), (2, 0), 32, 32, 32),
                                         ('enc l1 3x3 64>64 @128', 64, 64, (3, 3), (1, 1), 128, 128, 64)):
    x = torch.randn(n, cin, H, W, device=dev)
    w = torch.randn(cout, cin, *k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    pc = ops.PackedConv.from_weight(w, b, padding=pad)
    ref = ops.conv2d(pc, x, act=ops.ACT_RELU)
    w4 = ops.pack_conv_weight_a4(w, pc.g4, c4=True)[0]
    x4 = to_c4(x)
    out4 = torch.empty((n, cout // 4, H, W, 4), device=dev)
    run_c4(pc, w4, x4, out4, n, cin, H, W, ops.ACT_RELU)
    err = float((from_c4(out4, cout) - ref).abs().max())
    fl = 2.0 * cin * k[0] * k[1] * cout * H * W * n
    t0 = timeit(lambda: ops.conv2d(pc, x, out=ref, act=ops.ACT_RELU))
    t1 = timeit(lambda: run_c4(pc, w4, x4, out4, n, cin, H, W, ops.ACT_RELU))
    print(f'{name:26s} NCHW {t0:7.1f} us {fl / t0 / 1e6:6.1f} TF/s   NC/4HW4 {t1:7.1f} us {fl / t1 / 1e6:6.1f} TF/s   max|diff| {err:.2e}')
