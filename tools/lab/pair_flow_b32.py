"""r6: [corr_net.1 | flow_net.1] as one quarter-domain Winograd launch at batch N (fewer rounds of resident blocks) vs two launches.
    python tools/lab/pair_flow_b32.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from scflow_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model, _ = bench.build_model(8, 'cuda:0')
d = bench.make_batch(n, 1000, 'cuda:0')
for rep in range(3):
    for name, mode in (('never merged', 1), ('rule (default)', 0)):
        ops.tune('conv_pair', mode)
        for _ in range(3):
            out = bench.run_step(model, d)
        torch.cuda.synchronize()
        cs = f'{float(out[0][-1].double().abs().sum()):.9e}'
        t0 = time.perf_counter()
        for _ in range(30):
            bench.run_step(model, d)
        torch.cuda.synchronize()
        print(f'rep {rep} batch {n} {name:30s}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step  checksum {cs}', flush=True)
ops.tune('conv_pair', 0)
