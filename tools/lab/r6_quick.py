"""r6: single layers of the batch-32 step, Winograd dispatch vs direct kernels, launch-bound timers (conv_timing).
    python tools/lab/r6_quick.py [filter ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scflow_amd import ops
dev = 'cuda:0'
LAYERS = [  # name, cin, cout, k, stride, pad, H, W, N, bias, act
    ('menc1 64>32 3x3', 64, 32, (3, 3), 1, (1, 1), 32, 32, 32, True, ops.ACT_RELU),
    ('flow1 128>64 3x3', 128, 64, (3, 3), 1, (1, 1), 32, 32, 32, True, ops.ACT_RELU),
    ('ctx head 128>256 1x1 N32', 128, 256, (1, 1), 1, (0, 0), 32, 32, 32, True, ops.ACT_NONE),
    ('feat head 128>256 1x1 N64', 128, 256, (1, 1), 1, (0, 0), 32, 32, 64, True, ops.ACT_NONE),
    ('corr0 324>256 1x1', 324, 256, (1, 1), 1, (0, 0), 32, 32, 32, True, ops.ACT_RELU),
    ('flow0 2>128 7x7', 2, 128, (7, 7), 1, (3, 3), 32, 32, 32, True, ops.ACT_RELU),
    ('menc0 1>64 3x3', 1, 64, (3, 3), 1, (1, 1), 32, 32, 32, True, ops.ACT_RELU),
    ('fpred 256>2 3x3', 256, 2, (3, 3), 1, (1, 1), 32, 32, 32, True, ops.ACT_NONE),
    ('mpred 256>1 1x1', 256, 1, (1, 1), 1, (0, 0), 32, 32, 32, True, ops.ACT_SIGMOID),
    ('ds 64>96 1x1s2 N64', 64, 96, (1, 1), 2, (0, 0), 128, 128, 64, True, ops.ACT_NONE),
    ('ds 64>96 1x1s2 N32', 64, 96, (1, 1), 2, (0, 0), 128, 128, 32, True, ops.ACT_NONE),
    ('ds 96>128 1x1s2 N64', 96, 128, (1, 1), 2, (0, 0), 64, 64, 64, True, ops.ACT_NONE),
    ('ds 96>128 1x1s2 N32', 96, 128, (1, 1), 2, (0, 0), 64, 64, 32, True, ops.ACT_NONE),
    ('l1 64>64 3x3 @128 N64', 64, 64, (3, 3), 1, (1, 1), 128, 128, 64, True, ops.ACT_NONE),
    ('l1 64>64 3x3 @128 N32', 64, 64, (3, 3), 1, (1, 1), 128, 128, 32, True, ops.ACT_NONE),
    ('pose1 128>128 3x3s2 @16', 128, 128, (3, 3), 2, (1, 1), 16, 16, 32, False, ops.ACT_NONE),
    ('pose2 128>128 3x3s2 @8', 128, 128, (3, 3), 2, (1, 1), 8, 8, 32, False, ops.ACT_NONE),
]
sel = sys.argv[1:]
if not sel or 'ctxsplit' in sel:      # the context head as the refiner runs it: split tanh | relu epilogue into a slice of the GRU buffer
    x = torch.randn(32, 128, 32, 32, device=dev)
    w = torch.randn(256, 128, 1, 1, device=dev) * 0.05
    pc = ops.PackedConv.from_weight(w, torch.randn(256, device=dev), stride=1, padding=(0, 0))
    hx = torch.empty(32, 384, 32, 32, device=dev)
    dense = torch.empty(32, 256, 32, 32, device=dev)
    for tag, kw in (('dense out, no act', dict(out=dense)), ('dense out, tanh|relu', dict(out=dense, act=ops.ACT_TANH, act2=ops.ACT_RELU, act_split=128)),
                    ('slice of hx, no act', dict(out=hx[:, :256])), ('slice of hx, tanh|relu', dict(out=hx[:, :256], act=ops.ACT_TANH, act2=ops.ACT_RELU, act_split=128)),
                    ('slice of hx, relu', dict(out=hx[:, :256], act=ops.ACT_RELU))):
        for _ in range(3):
            ops.conv2d(pc, x, **kw)
        with ops.record_conv_kernels() as ran:
            ops.conv2d(pc, x, **kw)
        ops.conv_timing(True)
        for _ in range(21):
            ops.conv2d(pc, x, **kw)
        ts = sorted(e[0] for e in ops.conv_timing(False))
        print(f'ctx head 128>256 1x1 N32 {tag:26s}: {ts[len(ts) // 2]:7.1f} us [{ran[0][1]}]', flush=True)
for name, cin, cout, k, s, p, H, W, n, bias, act in LAYERS:
    if sel and not any(x in name for x in sel):
        continue
    x = torch.randn(n, cin, H, W, device=dev)
    w = torch.randn(cout, cin, *k, device=dev) * 0.05
    b = torch.randn(cout, device=dev) if bias else None
    pc = ops.PackedConv.from_weight(w, b, stride=s, padding=p)
    out = ops.conv2d(pc, x, act=act)
    flops = 2.0 * cin * k[0] * k[1] * cout * out.shape[2] * out.shape[3] * n
    res = []
    for wino in (True, False):
        prev = ops.set_conv_winograd(wino)
        for _ in range(3):
            ops.conv2d(pc, x, out=out, act=act)
        with ops.record_conv_kernels() as ran:
            ops.conv2d(pc, x, out=out, act=act)
        ops.conv_timing(True)
        for _ in range(21):
            ops.conv2d(pc, x, out=out, act=act)
        ts = sorted(e[0] for e in ops.conv_timing(False))
        us = ts[len(ts) // 2]
        res.append(f'{"auto" if wino else "direct"}: {us:7.1f} us {flops / us / 1e6:6.1f} TF/s [{ran[0][1] if ran else "?"}]')
        ops.set_conv_winograd(prev)
        if pc.wwino is None and pc.wwino1d is None:
            break
    print(f'{name:28s} ' + '   '.join(res), flush=True)
