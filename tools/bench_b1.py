import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
model, sd = bench.build_model(8, 'cuda:0')
b1 = bench.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 1, seed=5, device='cuda:0')
for _ in range(3):
    bench.run_step(model, b1)
torch.cuda.synchronize()
