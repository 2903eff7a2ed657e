"""r6: is the hipGraph replay of the batch-N pass bound by the HOST (hipGraphLaunch walks the graph and enqueues every node) or by
the GPU?  Times (a) the host loop that issues K replays, until the last graph.replay() returns, (b) the same until the GPU is done.
    python tools/lab/graph_host_cost.py [batch] [replays]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from scflow_amd.graph import GraphedRefiner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
model, _ = bench.build_model(8, 'cuda:0')
d = bench.make_batch(n, 5, 'cuda:0')
g = GraphedRefiner(model, d)
for k in g.static_in:
    g.static_in[k].copy_(d[k])
for _ in range(5):
    g()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K):
        g.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'batch {n}: {K} replays issued in {(t1 - t0) / K * 1e3:.3f} ms each (host), done after {(t2 - t0) / K * 1e3:.3f} ms each '
          f'(host + GPU); the GPU was {(t2 - t1) * 1e3:.2f} ms behind the host at the end', flush=True)
# one replay at a time (latency of a single pass, queues empty before it)
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    ts.append((t1 - t0, time.perf_counter() - t0))
ts.sort(key=lambda a: a[1])
print(f'batch {n}: ONE replay from idle: replay() returns after {ts[10][0] * 1e3:.3f} ms, GPU done after {ts[10][1] * 1e3:.3f} ms (median of 20)')
# eager for comparison
for _ in range(3):
    bench.run_step(model, d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    bench.run_step(model, d)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'batch {n}: eager: {K} passes issued in {(t1 - t0) / K * 1e3:.3f} ms each (host), done after {(t2 - t0) / K * 1e3:.3f} ms each')
