"""Micro-benchmark of single convolution layers of the SCFlow path (HIP events, median)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scflow_amd import ops
dev = 'cuda:0'
N = int(os.environ.get('N', 32))
LAYERS = [  # name, cin, cout, (kh,kw), stride, pad, H, W, batch multiplier
    ('heads 3x3 128>512', 128, 512, (3, 3), 1, (1, 1), 32, 32, 1),
    ('corr_net.1 3x3 256>192', 256, 192, (3, 3), 1, (1, 1), 32, 32, 1),
    ('gru zr 1x5 384>256', 384, 256, (1, 5), 1, (0, 2), 32, 32, 1),
    ('gru q 5x1 384>128', 384, 128, (5, 1), 1, (2, 0), 32, 32, 1),
    ('enc l1 3x3 64>64 @128', 64, 64, (3, 3), 1, (1, 1), 128, 128, 2),
    ('enc l2 3x3 96>96 @64', 96, 96, (3, 3), 1, (1, 1), 64, 64, 2),
    ('1x1 324>256', 324, 256, (1, 1), 1, (0, 0), 32, 32, 1),
    ('pose c0 3x3s2 224>128', 224, 128, (3, 3), 2, (1, 1), 32, 32, 1),
    ('flow_pred 3x3 256>2', 256, 2, (3, 3), 1, (1, 1), 32, 32, 1),
    ('stem 7x7s2 3>64 @256', 3, 64, (7, 7), 2, (3, 3), 256, 256, 2),
    ('flow 7x7 2>128', 2, 128, (7, 7), 1, (3, 3), 32, 32, 1),
    ('mask 3x3 1>64', 1, 64, (3, 3), 1, (1, 1), 32, 32, 1),
    ('pose c1 3x3s2 128>128 @16', 128, 128, (3, 3), 2, (1, 1), 16, 16, 1),
    ('pose c2 3x3s2 128>128 @8', 128, 128, (3, 3), 2, (1, 1), 8, 8, 1),
    ('enc 3x3 128>64', 128, 64, (3, 3), 1, (1, 1), 32, 32, 1),
    ('menc 3x3 64>32', 64, 32, (3, 3), 1, (1, 1), 32, 32, 1),
    ('ds 1x1s2 64>96 @128', 64, 96, (1, 1), 2, (0, 0), 128, 128, 2),
]
sel = sys.argv[1:] 
for name, cin, cout, k, s, p, H, W, bm in LAYERS:
    if sel and not any(x in name for x in sel):
        continue
    n = N * bm
    x = torch.randn(n, cin, H, W, device=dev)
    w = torch.randn(cout, cin, *k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    pc = ops.PackedConv.from_weight(w, b, stride=s, padding=p)
    out = ops.conv2d(pc, x, act=ops.ACT_RELU)
    flops = 2.0 * cin * k[0] * k[1] * cout * out.shape[2] * out.shape[3] * n
    res = []
    if pc.wtaps is not None:        # thin input: A/B of the tap-contracting kernel against the chunked one,
        import copy                 # kernel durations from launch-bound timers
        for tag, q in (('taps', pc), ('chunked', copy.copy(pc))):
            if tag == 'chunked':
                q.wtaps, q.desc = None, None
            for _ in range(3):
                ops.conv2d(q, x, out=out, act=ops.ACT_RELU)
            ops.conv_timing(True)
            for _ in range(15):
                ops.conv2d(q, x, out=out, act=ops.ACT_RELU)
            ts = sorted(e[0] for e in ops.conv_timing(False))
            res.append(f'{tag}: {ts[len(ts) // 2]:7.1f} us kernel')
    for prec in (('f32', 'f16x3') if not os.environ.get('F32_ONLY') else ('f32',)):
        ops.set_conv_precision(prec)
        for _ in range(3):
            ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)
        evs = []
        for _ in range(15):
            a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            a.record(); ops.conv2d(pc, x, out=out, act=ops.ACT_RELU); e.record(); evs.append((a, e))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(e) for a, e in evs)
        us = ts[len(ts) // 2] * 1e3
        res.append(f'{prec}: {us:7.1f} us {flops / us / 1e6:6.1f} TF/s')
    ops.set_conv_precision('f32')
    print(f'{name:26s} N={n:3d}  ' + '   '.join(res))
