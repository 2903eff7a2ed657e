"""Lab: pose-head linear layers alone (fc1 2048->1024, fc2 1024->256, rotation|translation pair 256->126|63)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops
DEV = 'cuda:0'
def bench(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = sorted(ops.time_first_kernel(fn) for _ in range(reps))
    return ts[len(ts) // 2]
for n in (32, 1):
    out = []
    for k, o in ((2048, 1024), (1024, 256)):
        x = torch.randn((n, k), device=DEV); w = torch.randn((o, k), device=DEV) * k ** -0.5; b = torch.randn((o,), device=DEV)
        out.append(f'{k}->{o}: {bench(lambda: ops.linear(x, w, b, ops.ACT_RELU)):.1f} us')
    x = torch.randn((n, 256), device=DEV)
    w1, w2 = torch.randn((126, 256), device=DEV), torch.randn((63, 256), device=DEV)
    out.append(f'pair 256->126|63: {bench(lambda: ops.linear_pair(x, w1, None, w2, None)):.1f} us')
    print(f'N={n}:', '  '.join(out))
