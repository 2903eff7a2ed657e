"""Lab: cost of the GRU z|r launch with / without the pre-activation context term (res), and of the
plain convolution of the same shape: isolates the epilogue."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops
DEV = 'cuda:0'
n, H, W = 32, 32, 32
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = [ops.time_first_kernel(fn) for _ in range(reps)]
    ts.sort()
    return ts[len(ts) // 2]
for (k, pad) in (((1, 5), (0, 2)), ((5, 1), (2, 0))):
    for cin in (128, 256, 384):        # three chain lengths: the intercept is the launch's fixed cost
        hx = torch.randn((n, 384, H, W), device=DEV)
        w = torch.randn((256, cin, *k), device=DEV) * 0.05
        b = torch.randn((256,), device=DEV)
        pc = ops.PackedConv.from_weight(w, b, padding=pad)
        pcn = ops.PackedConv.from_weight(w, None, padding=pad)
        z = torch.empty((n, 128, H, W), device=DEV); rh = torch.empty_like(z)
        out = torch.empty((n, 256, H, W), device=DEV)
        ctx = torch.randn((n, 384, H, W), device=DEV)
        hv = hx[:, :128]
        if cin == 384:
            x0, x1 = hx, None
        elif cin == 128:
            x0, x1 = hx[:, :128], None
        else:
            x0, x1 = hx[:, :128], hx[:, 256:]
        fl = 2.0 * n * 256 * cin * 5 * H * W
        t_plain = bench(lambda: ops.conv2d(pc, x0, x1, out=out))
        t_zr = bench(lambda: ops.conv2d(pc, x0, x1, out=z, mode=ops.CONV_GRU_ZR, gru_h=hv, gru_aux=rh))
        t_zr_res = bench(lambda: ops.conv2d(pcn, x0, x1, out=z, mode=ops.CONV_GRU_ZR, gru_h=hv, gru_aux=rh, res=ctx[:, :256]))
        print(f'{cin}->256 {k}: plain {t_plain:.1f} us ({fl / t_plain * 1e-6:.1f} TF)  GRU z|r {t_zr:.1f} us ({fl / t_zr * 1e-6:.1f} TF)  '
              f'GRU z|r + ctx {t_zr_res:.1f} us ({fl / t_zr_res * 1e-6:.1f} TF)')
