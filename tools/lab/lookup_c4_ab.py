"""configs[4] (RAFTRefinerFlowMask, 8 x 480x640, 12 iterations): the lookup's groups-per-block choice A/B'd inside the step.
    python tools/lab/lookup_c4_ab.py [modes, e.g. 1,0,6]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch

import bench
from scflow_amd import ops


def main():
    modes = [int(m) for m in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['1', '0'])]
    # bench.config4_block builds the model and times it; here the same step with the knob alternating
    import scflow_amd
    m = scflow_amd.build_refiner(scflow_amd.raft_model_cfg(iters=12))
    m.load_state_dict(scflow_amd.fill_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=9), strict=True)
    m = m.to('cuda')
    g = torch.Generator().manual_seed(3)
    a = torch.rand((8, 3, 480, 640), generator=g).cuda()
    b = torch.rand((8, 3, 480, 640), generator=g).cuda()
    for _ in range(3):
        m.get_flow(a, b)
    torch.cuda.synchronize()
    ops.lookup_timing(True, reserve=12 * 16)
    for rep in range(3):
        for mode in modes:
            ops.tune('lookup_pipe', mode)
            m.get_flow(a, b)
            torch.cuda.synchronize()
            ops.lookup_timing_reset()
            for _ in range(5):
                m.get_flow(a, b)
            torch.cuda.synchronize()
            us = ops.lookup_timing_read()
            q = 8 * 60 * 80
            print(f'rep {rep} lookup_pipe={mode}: mean {statistics.fmean(us):6.2f} us median {statistics.median(us):6.2f} '
                  f'frac {2904.0 * q / statistics.fmean(us) / 1e3 / 8000:.3f}', flush=True)
    ops.tune('lookup_pipe', 0)
    ops.lookup_timing(False)


if __name__ == '__main__':
    main()
