"""Per-layer convolution times of ONE batch-N get_pose (launch-bound timers, Python-sequenced loop):
where a small-batch pass spends its kernel time.   python tools/lab/b1_layers.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from scflow_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model, _ = bench.build_model(8, 'cuda:0')
b = bench.make_batch(n, 5, 'cuda:0')
for _ in range(3):
    bench.run_step(model, b)
ops.conv_timing(True)
bench.run_step(model, b)
ev = ops.conv_timing(False)
agg = {}
for us, fl, tag in ev:
    a = agg.setdefault(tag, [0, 0.0, 0.0]); a[0] += 1; a[1] += us; a[2] += fl
tot = sum(v[1] for v in agg.values())
print(f'batch {n}: {len(ev)} conv launches, {tot:.0f} us of kernel time')
for tag, (c, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'  {tag:40s} x{c:3d} {us:8.1f} us  avg {us / c:6.1f}  {fl / us / 1e6:6.1f} TF/s')
