"""r6 (costed on measurements): what would 'conv_q,x(motion) computed inside the z|r launch, q contracting
only r.h' cost?  F(4, 5) launches of the shapes involved, plain bias epilogue (the gate epilogues cost the same or less,
notebook 3.2d), batch 32, 32 x 32 maps, launch-bound timers:
    z|r today        K = 256 -> 256   512 blocks x 64 chunks
    q today          K = 256 -> 128   256 blocks x 64 chunks
    z|r + q_x        <= K = 256 -> 384 (768 blocks x 64 chunks; the merged launch would run 256 of them for 32 chunks)
                     >= K = 128 -> 384 (768 blocks x 32 chunks) + half of (z|r today)
    q over r.h only  K = 128 -> 128   256 blocks x 32 chunks
    python tools/lab/r6_gru_bound.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scflow_amd import ops
dev = 'cuda:0'
N = int(os.environ.get('N', 32))
for k, pad, tag in (((1, 5), (0, 2), '1x5'), ((5, 1), (2, 0), '5x1')):
    row = {}
    for cin, cout in ((256, 256), (256, 128), (256, 384), (128, 384), (128, 128), (128, 256), (384, 256), (384, 128)):
        x = torch.randn(N, cin, 32, 32, device=dev)
        w = torch.randn(cout, cin, *k, device=dev) * (1.0 / (cin * 5)) ** 0.5
        pc = ops.PackedConv.from_weight(w, torch.randn(cout, device=dev) * 0.1, stride=1, padding=pad)
        out = torch.empty(N, cout, 32, 32, device=dev)
        for _ in range(50):
            ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)
        with ops.record_conv_kernels() as ran:
            ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)
        ops.conv_timing(True)
        for _ in range(31):
            ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)
        ts = sorted(e[0] for e in ops.conv_timing(False))
        us = ts[len(ts) // 2]
        row[(cin, cout)] = us
        fl = 2.0 * cin * 5 * cout * 1024 * N * 0.4
        print(f'{tag} {cin:3d}->{cout:3d}  blocks {N * 1024 // 256 * (cout // 64):4d} x {cin // 4:2d} chunks: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s executed  [{ran[0][1]}]', flush=True)
    today = row[(256, 256)] + row[(256, 128)]
    lo = row[(128, 384)] + 0.5 * row[(256, 256)] + row[(128, 128)]
    hi = row[(256, 384)] + row[(128, 128)]
    print(f'{tag}: today z|r + q = {today:.1f} us;  z|r+q_x & q(r.h): between {lo:.1f} and {hi:.1f} us', flush=True)
