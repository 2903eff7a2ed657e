"""Step time with one library build: SCF_EXP_SUFFIX selects tools/lab/bin/libscflow_hip_exp<suffix>.so (unset: the product
library).  Run once per build, alternating, on one box:   [SCF_EXP_SUFFIX=_x] python tools/lab/lib_ab.py [batch] [steps] [reps]
GRAPH=1: hipGraph replays (small batches)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
if os.environ.get('SCF_EXP_SUFFIX'):
    from scflow_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', f'libscflow_hip_exp{os.environ["SCF_EXP_SUFFIX"]}.so')
import torch

import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
model, _ = bench.build_model(8, 'cuda')
d = bench.make_batch(batch, 1000, 'cuda')
if os.environ.get('GRAPH') == '1':
    from scflow_amd.graph import GraphedRefiner
    g = GraphedRefiner(model, d)
    step = lambda: g(d)
else:
    step = lambda: bench.run_step(model, d)
# a timing is only worth something if the build computes the same thing: checksum of the last iteration's flow and pose
out = bench.run_step(model, d)
torch.cuda.synchronize()
print(f'lib {os.environ.get("SCF_EXP_SUFFIX", "(product)"):10s} checksum: flow {float(out[0][-1].double().abs().sum()):.6e} rot {float(out[2][-1].double().sum()):.9f} '
      f'finite {bool(torch.isfinite(out[0][-1]).all())}', flush=True)
for rep in range(reps):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'lib {os.environ.get("SCF_EXP_SUFFIX", "(product)"):10s} rep {rep}: {dt * 1e3:7.3f} ms per step  {batch / dt:7.1f} pairs/s', flush=True)
