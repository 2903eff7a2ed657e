"""The direct-kernel layers of the batch-32 step (bench.py's top_layers without a Winograd form): launch time and the
tile the dispatch picks, under the K-split knobs of scf_tune."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops
DEV = 'cuda:0'
# n, cin, cout, k, stride, pad, H, W (input size)
cases = [('324->256 1x1 @32 N32', 32, 324, 256, 1, 1, 0, 32, 32), ('224->128 3x3/s2 @32->16 N32', 32, 224, 128, 3, 2, 1, 32, 32),
         ('64->96 3x3/s2 @128->64 N64', 64, 64, 96, 3, 2, 1, 128, 128), ('96->128 3x3/s2 @64->32 N64', 64, 96, 128, 3, 2, 1, 64, 64),
         ('128->128 3x3/s2 @16->8 N32', 32, 128, 128, 3, 2, 1, 16, 16), ('128->128 3x3/s2 @8->4 N32', 32, 128, 128, 3, 2, 1, 8, 8),
         ('128->256 1x1 @32 N32', 32, 128, 256, 1, 1, 0, 32, 32), ('64->96 1x1/s2 @128->64 N64', 64, 64, 96, 1, 2, 0, 128, 128)]
lib = ops._lib.load()
torch.manual_seed(0)
for name, n, cin, cout, k, s, pad, H, W in cases:
    x = torch.randn((n, cin, H, W), device=DEV)
    w = torch.randn((cout, cin, k, k), device=DEV) * (1.0 / (cin * k * k)) ** 0.5
    b = torch.randn((cout,), device=DEV) * 0.1
    pc = ops.PackedConv.from_weight(w, b, stride=s, padding=pad)
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    fl = 2.0 * n * cout * cin * k * k * Ho * Wo
    row = []
    for fk, gg in ((0, 0), (1, 0), (1, 1)):
        ops.tune('dma_force_ksplit', fk); ops.tune('dma_ksplit_groups', gg)
        d, out = ops.conv_desc(pc, x, act=ops.ACT_RELU)
        info = (C.c_int32 * 4)()
        lib.scf_conv2d_query(C.byref(d), info)
        with ops.record_conv_kernels() as ran:
            ops.conv2d(pc, x, act=ops.ACT_RELU)
        for _ in range(20):
            ops.conv2d(pc, x, act=ops.ACT_RELU)
        ts = sorted(ops.time_first_kernel(lambda: ops.conv2d(pc, x, act=ops.ACT_RELU)) for _ in range(9))
        row.append(f'{ran[0][1]} {list(info)} {ts[4]:6.1f} us {fl / ts[4] * 1e-6:5.1f} TF')
    ops.tune('dma_force_ksplit', 0); ops.tune('dma_ksplit_groups', 0)
    print(f'{name:30s} ' + ' | '.join(row), flush=True)
