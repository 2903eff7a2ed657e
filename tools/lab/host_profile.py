"""Where the HOST time of an eager batch-1 get_pose goes (cProfile, GPU box): the pass is bound by
Python / ctypes launch overhead, not by the GPU."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

model, _ = bench.build_model(8, 'cuda:0')
b1 = bench.make_batch(1, 5, 'cuda:0')
for _ in range(5):
    bench.run_step(model, b1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    bench.run_step(model, b1)
torch.cuda.synchronize()
print(f'eager: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per pair')
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    bench.run_step(model, b1)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
