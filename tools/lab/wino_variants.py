"""F(2x2, 3x3) kernel variants (scf_tune wino_variant: 1 = pair kernel, 2 / 3 = quarter-domain kernel with 4 / 8
waves): (a) every variant against the direct kernel on the stress shapes (two segments, BN + residual, ragged,
dword patches), (b) launch-bound kernel times per layer shape of the batch-32 step and of configs[4].
    python tools/lab/wino_variants.py [check|time] [shape filter ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get('SCF_EXP_SUFFIX') is not None:
    from scflow_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', f'libscflow_hip_exp{os.environ["SCF_EXP_SUFFIX"]}.so')
from scflow_amd import ops
DEV = 'cuda:0'
VARIANTS = [int(v) for v in os.environ.get('SCF_VARIANTS', '1,2,3').split(',')]


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def check():
    cases = [(16, 64, 64, 32, 32, 0, False, False, True), (2, 128, 512, 32, 32, 0, False, False, True),
             (8, 256, 126, 32, 32, 192, False, False, True), (3, 96, 96, 64, 64, 0, True, True, True),
             (1, 64, 64, 128, 128, 0, False, False, False), (24, 30, 40, 20, 28, 0, False, True, False),
             (4, 128, 64, 60, 80, 0, False, False, True), (128, 16, 32, 7, 10, 0, False, False, True),
             (64, 16, 64, 16, 16, 0, False, False, True), (5, 64, 96, 64, 64, 0, False, True, True),
             (40, 32, 64, 48, 32, 0, False, False, True), (6, 64, 128, 33, 40, 16, True, True, True),
             (32, 20, 64, 23, 30, 0, False, False, True), (64, 64, 64, 8, 8, 0, False, False, True)]
    bad = 0
    for case in cases:
        n, cin, cout, H, W, c0, bn, with_res, relu = case
        x = rnd((n, cin, H, W), 40).to(DEV)
        wt = rnd((cout, cin, 3, 3), 41, (1.0 / (cin * 9)) ** 0.5)
        b = rnd((cout,), 42, 0.1)
        bnp = None
        if bn:
            bnp = [t.to(DEV) for t in (rnd((cout,), 43) * 0.2 + 1, rnd((cout,), 44) * 0.1, rnd((cout,), 45) * 0.1,
                                       rnd((cout,), 46).abs() * 0.5 + 0.5)]
        pc = ops.PackedConv.from_weight(wt.to(DEV), b.to(DEV), padding=1, bn=bnp)
        res = rnd((n, cout, H, W), 47).to(DEV) if with_res else None
        kw = dict(res=res, act=ops.ACT_RELU if relu else ops.ACT_NONE)
        x0, x1 = (x[:, :c0], x[:, c0:]) if c0 else (x, None)
        ops.set_conv_winograd(False)
        want = ops.conv2d(pc, x0, x1, **kw)
        ops.set_conv_winograd(True)
        line = f'{str(case):60s}'
        for v in VARIANTS:
            ops.tune('wino_variant', v)
            with ops.record_conv_kernels() as ran:
                got = ops.conv2d(pc, x0, x1, **kw)
            torch.cuda.synchronize()
            err = float((got - want).abs().max())
            import ctypes as C
            d, _ = ops.conv_desc(pc, x0, x1, **kw)
            info = (C.c_int32 * 4)()
            ops._lib.load().scf_conv2d_query(C.byref(d), info)
            ok = err <= 3e-5 and bool(torch.isfinite(got).all())
            bad += 0 if ok else 1
            line += f' | v{v}: {ran[0][1][:8]:8s} frags/blk {info[1]} blocks {info[2]:5d} err {err:.1e}{"" if ok else " FAIL"}'
        ops.tune('wino_variant', 0)
        print(line, flush=True)
    print('CHECK', 'FAILED' if bad else 'ok', bad)
    return bad


def time_layers(filters):
    cases = [('64->64 @128 N64', 64, 64, 64, 128, 128), ('64->64 @128 N32', 32, 64, 64, 128, 128),
             ('96->96 @64 N64', 64, 96, 96, 64, 64), ('128->128 @32 N64', 64, 128, 128, 32, 32),
             ('128->512 @32 N32', 32, 128, 512, 32, 32), ('256->192 @32 N32', 32, 256, 192, 32, 32),
             ('256->126 @32 N32', 32, 256, 126, 32, 32), ('128->64 @32 N32', 32, 128, 64, 32, 32),
             ('64->32 @32 N32', 32, 64, 32, 32, 32),
             ('128->512 @60x80 N8', 8, 128, 512, 60, 80), ('256->192 @60x80 N8', 8, 256, 192, 60, 80),
             ('128->512 @32 N4', 4, 128, 512, 32, 32)]
    if filters:
        cases = [c for c in cases if any(a in c[0] for a in filters)]
    ops.set_conv_winograd(True)
    for name, n, cin, cout, H, W in cases:
        x = torch.randn((n, cin, H, W), device=DEV)
        w = torch.randn((cout, cin, 3, 3), device=DEV) * (1.0 / (cin * 9)) ** 0.5
        b = torch.randn((cout,), device=DEV) * 0.1
        pc = ops.PackedConv.from_weight(w, b, padding=1)
        out = torch.empty((n, cout, H, W), device=DEV)
        fl = 2.0 * n * cout * cin * 9 * H * W / 2.25          # executed flops
        line = f'{name:22s}'
        for v in VARIANTS:
            ops.tune('wino_variant', v)
            for _ in range(200):          # the shader clock takes a while to settle
                ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)
            ts = sorted(ops.time_first_kernel(lambda: ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)) for _ in range(9))
            line += f' | v{v} {ts[4]:7.1f} us {fl / ts[4] * 1e-6:6.1f} TF/s exec'
        ops.tune('wino_variant', 0)
        print(line, flush=True)


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'check'
    if mode == 'check':
        sys.exit(1 if check() else 0)
    time_layers(sys.argv[2:])
