"""Lab: per-layer convolution table of one configs[4] step (RAFTRefinerFlowMask, 8 x 480x640, 12 iterations)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scflow_amd
from scflow_amd import ops
dev = 'cuda:0'
n, H, W, iters = 8, 480, 640, 12
m = scflow_amd.build_refiner(scflow_amd.raft_model_cfg(iters=iters))
sd = scflow_amd.fill_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=9)
m.load_state_dict(sd, strict=True)
m = m.to(dev)
g = torch.Generator().manual_seed(3)
rend = torch.rand((n, 3, H, W), generator=g).to(dev)
real = torch.rand((n, 3, H, W), generator=g).to(dev)
for _ in range(2):
    m.get_flow(rend, real)
torch.cuda.synchronize()
ops.conv_timing(True)
m.get_flow(rend, real)
ev = ops.conv_timing(False)
by = {}
for us, fl, tag in ev:
    a = by.setdefault(tag, [0, 0.0, 0.0]); a[0] += 1; a[1] += us; a[2] += fl
tot_us = sum(v[1] for v in by.values()); tot_fl = sum(v[2] for v in by.values())
print(f'conv total {tot_us:.0f} us, {tot_fl / tot_us / 1e6:.1f} TF/s')
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f'  {k:44s} x{v[0]:3d} {v[1]:9.1f} us {v[2] / v[1] / 1e6:6.1f} TF')
