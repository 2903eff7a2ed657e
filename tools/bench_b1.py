"""Batch-1 (BASELINE configs[1]) latency: eager vs hipGraph replay; `python tools/bench_b1.py [N]`."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scflow_amd.graph import GraphedRefiner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model, sd = bench.build_model(8, 'cuda:0')
b1 = bench.make_batch(n, seed=5, device='cuda:0')
for _ in range(3):
    bench.run_step(model, b1)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    bench.run_step(model, b1)
torch.cuda.synchronize()
print(f'batch {n}: eager {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per step')
g = GraphedRefiner(model, b1)
for _ in range(3):
    g(b1)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    g(b1)
torch.cuda.synchronize()
print(f'batch {n}: hipGraph {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per step')
