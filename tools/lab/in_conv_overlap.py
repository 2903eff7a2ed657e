"""r6: do an InstanceNorm pass (HBM-bound) and a Winograd convolution (MFMA-bound) hide each other?  Two streams, launch-bound wall time;
"ONE LAUNCH" needs the lab entry point scf_conv2d_instance_norm (conv_wino_q_in_kernel: built, measured 387 vs 299 us, removed -- notebook
R6.10) and is skipped without it.
    python tools/lab/in_conv_overlap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scflow_amd import ops
dev = 'cuda:0'
for (nc, c, hw, nn) in ((32, 64, 128, 64), (32, 96, 64, 64), (32, 128, 32, 64)):
    x = torch.randn(nc, c, hw, hw, device=dev)
    w = torch.randn(c, c, 3, 3, device=dev) * 0.05
    pc = ops.PackedConv.from_weight(w, torch.randn(c, device=dev), stride=1, padding=(1, 1))
    y = torch.empty_like(x)
    t = torch.randn(nn, c, hw, hw, device=dev)
    s2 = torch.cuda.Stream()
    def conv(): ops.conv2d(pc, x, out=y)
    def inorm(): ops.instance_norm(t, relu=True, out=t)
    def timeit(fn, reps=20):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    def both():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(s2):
            s2.wait_event(ev)
            inorm()
        conv()
        torch.cuda.current_stream().wait_stream(s2)
    def serial():
        conv(); inorm()
    a, b, sser, par = timeit(conv), timeit(inorm), timeit(serial), timeit(both)
    print(f'{c}->{c} @{hw} conv N{nc}: {a:7.1f} us | IN N{nn}: {b:7.1f} us | in order {sser:7.1f} us | two streams {par:7.1f} us', flush=True)
