"""r6: every convolution launch of one configs[4] step (RAFTRefinerFlowMask, 8 x 480x640, 12 iterations) with its launch-bound time:
where that step's 25.7 ms go.   python tools/lab/c4_layers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import scflow_amd
from scflow_amd import ops
n, H, W, iters = 8, 480, 640, 12
m = scflow_amd.build_refiner(scflow_amd.raft_model_cfg(iters=iters))
sd = scflow_amd.fill_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=9)
m.load_state_dict(sd, strict=True)
m = m.to('cuda:0')
g = torch.Generator().manual_seed(3)
rend = torch.rand((n, 3, H, W), generator=g).to('cuda:0')
real = torch.rand((n, 3, H, W), generator=g).to('cuda:0')
for _ in range(2):
    m.get_flow(rend, real)
torch.cuda.synchronize()
runs = []
for _ in range(3):
    ops.conv_timing(True)
    m.get_flow(rend, real)
    runs.append(ops.conv_timing(False))
ev = [(sorted(r[i][0] for r in runs)[1], runs[0][i][1], runs[0][i][2]) for i in range(len(runs[0]))]
agg = {}
for us, fl, tag in ev:
    a = agg.setdefault(tag, [0, 0.0, 0.0]); a[0] += 1; a[1] += us; a[2] += fl
tot = sum(e[0] for e in ev)
print(f'{len(ev)} conv launches, {tot / 1e3:.2f} ms of kernel time')
for tag, (c, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    div = 2.25 if tag.endswith('[winograd]') else 2.5 if tag.endswith('F(4,5)]') else (1 / 0.6) if tag.endswith('F(2,5)]') else 1.0
    print(f'  {tag:52s} x{c:3d} {us:9.1f} us  avg {us / c:7.1f}  {fl / us / 1e6 / div:6.1f} TF/s executed ({fl / us / 1e6 / div / 157.3:.2f})')
