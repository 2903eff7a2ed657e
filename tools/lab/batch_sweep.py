"""r6: the whole refiner against the oracle over batch sizes BETWEEN the ones the tests pin (1, 2, 3, 32): which convolution
kernel takes a layer depends on the grid size, so every batch size is its own dispatch plan.  2 iterations, per-pair EPE.
    python tools/lab/batch_sweep.py [sizes ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, oracle, scflow_amd
from scflow_amd import ops


def run(sizes, iters=2, verbose=True):
    model, sd = bench.build_model(iters, 'cuda:0')
    torch.set_num_threads(bench.host_cores())
    worst_all = 0.0
    for n in sizes:
        inp = scflow_amd.make_inputs(n, 256, 256, seed=300 + n)
        d = {k: v.to('cuda:0') for k, v in inp.items()}
        with ops.record_conv_kernels() as ran:
            got = model.get_pose(d['render_images'], d['real_images'], d['ref_rotation'], d['ref_translation'], d['depth'],
                                 d['internel_k'], d['label'])
        torch.cuda.synchronize()
        with torch.no_grad():
            want = oracle.get_pose(inp['render_images'], inp['real_images'], inp['ref_rotation'], inp['ref_translation'],
                                   inp['depth'], inp['internel_k'], inp['label'], sd, iters=iters)
        valid = inp['depth'] > 0
        worst = 0.0
        for it in range(iters):
            for s_ in range(n):
                worst = max(worst, oracle.end_point_error(got[0][it][s_:s_ + 1].cpu(), want[0][it][s_:s_ + 1], valid[s_:s_ + 1]),
                            oracle.end_point_error(got[1][it][s_:s_ + 1].cpu(), want[1][it][s_:s_ + 1]))
        er = float((got[2][-1].cpu() - want[2][-1]).abs().max())
        fam = {}
        for _, k in ran:
            fam[k] = fam.get(k, 0) + 1
        worst_all = max(worst_all, worst)
        if verbose:
            print(f'batch {n:3d}: worst per-pair EPE {worst:.2e} px, max |dR| {er:.2e}  kernels {fam}', flush=True)
        assert worst <= 1e-3 and er <= 2e-5, (n, worst, er)
    return worst_all


if __name__ == '__main__':
    sizes = [int(a) for a in sys.argv[1:]] or [4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 28, 40, 48, 64]
    print('worst', run(sizes))
