"""Lab: the GRU q launch (two input segments, tanh + state update epilogue, context term) on the F(4, 5) kernel,
1x5 vs 5x1, against the plain launch of the same shape: what the q epilogue costs per pass direction."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops
DEV = 'cuda:0'
n, H, W = 32, 32, 32
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = sorted(ops.time_first_kernel(fn) for _ in range(reps))
    return ts[len(ts) // 2]
for (k, pad) in (((1, 5), (0, 2)), ((5, 1), (2, 0))):
    hx = torch.randn((n, 384, H, W), device=DEV)
    rh = torch.randn((n, 128, H, W), device=DEV)
    z = torch.rand((n, 128, H, W), device=DEV)
    w = torch.randn((128, 256, *k), device=DEV) * 0.05
    b = torch.randn((128,), device=DEV)
    pc, pcn = ops.PackedConv.from_weight(w, b, padding=pad), ops.PackedConv.from_weight(w, None, padding=pad)
    ctx = torch.randn((n, 128, H, W), device=DEV)
    out = torch.empty((n, 128, H, W), device=DEV)
    hv, xv = hx[:, :128], hx[:, 256:]
    t_plain = bench(lambda: ops.conv2d(pc, rh, xv, out=out))
    t_q = bench(lambda: ops.conv2d(pc, rh, xv, out=hv, mode=ops.CONV_GRU_Q, gru_h=hv, gru_z=z))
    t_q_sep = bench(lambda: ops.conv2d(pc, rh, xv, out=out, mode=ops.CONV_GRU_Q, gru_h=hv, gru_z=z))
    t_q_res = bench(lambda: ops.conv2d(pcn, rh, xv, out=hv, mode=ops.CONV_GRU_Q, gru_h=hv, gru_z=z, res=ctx))
    print(f'q 256->128 {k}: plain {t_plain:.1f} us  q in place {t_q:.1f}  q to another buffer {t_q_sep:.1f}  q + ctx in place {t_q_res:.1f}')
