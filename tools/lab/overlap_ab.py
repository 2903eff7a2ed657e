"""Lab: which decoder branches pay on a second stream at batch 32 (ops.small_work decides per branch)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from scflow_amd import ops
dev = 'cuda:0'
model, _ = bench.build_model(8, dev)
batch = bench.make_batch(32, seed=1, device=dev)
orig = ops.small_work
def run(tag, branches):
    ops.small_work = (lambda n, h, w, branch=None: branch in branches) if branches is not None else orig
    for _ in range(5):
        bench.run_step(model, batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(10):
            bench.run_step(model, batch)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10)
    ts.sort()
    print(f'{tag:28s} {ts[1] * 1e3:.3f} ms/step  {32 / ts[1]:.1f} pairs/s')
run('no overlap (product)', None)
run('mask', {'mask'})
run('upsample', {'upsample'})
run('flow', {'flow'})
run('mask + upsample', {'mask', 'upsample'})
run('no overlap again', None)
