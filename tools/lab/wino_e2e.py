"""End to end with the Winograd kernel on the 3x3 stride-1 layers: flow EPE vs the CPU oracle (2 pairs, 8
iterations), the golden outputs' error, and the batch-32 step time -- each next to the direct kernels."""
import os, sys, json, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import scflow_amd, oracle, bench
from scflow_amd import ops
DEV = 'cuda:0'
gd = os.path.join(ROOT, 'tests', 'golden')
shapes = json.load(open(os.path.join(gd, 'state_dict_keys.json')))['shapes']
sd = scflow_amd.fill_state_dict(shapes, seed=0)
m = scflow_amd.build_refiner(scflow_amd.scflow_model_cfg())
m.load_state_dict(sd, strict=True)
m = m.to(DEV)
inp = scflow_amd.make_inputs(2, 256, 256, seed=13)
torch.set_num_threads(bench.host_cores())
with torch.no_grad():
    want = oracle.get_pose(inp['render_images'], inp['real_images'], inp['ref_rotation'], inp['ref_translation'],
                           inp['depth'], inp['internel_k'], inp['label'], sd, iters=8)
d = {k: v.to(DEV) for k, v in inp.items()}
valid = inp['depth'] > 0
g = np.load(os.path.join(gd, 'refiner_full.npz'))
ginp = {k: v.to(DEV) for k, v in scflow_amd.make_inputs(int(g['n']), 256, 256, seed=int(g['input_seed'])).items()}
names = ['flow_from_pose', 'flow_from_pred', 'rotation', 'translation', 'mask', 'delta_rotation', 'delta_translation']
for wino in (False, True):
    ops.set_conv_winograd(wino)
    m.decoder.iters = 8
    got = m.get_pose(d['render_images'], d['real_images'], d['ref_rotation'], d['ref_translation'], d['depth'],
                     d['internel_k'], d['label'])
    worst = 0.0
    for it in range(8):
        worst = max(worst, oracle.end_point_error(got[0][it].cpu(), want[0][it], valid),
                    oracle.end_point_error(got[1][it].cpu(), want[1][it]))
    print(f'winograd={wino}: worst EPE over 8 iterations vs the CPU oracle {worst:.2e} px; final rotation err '
          f'{float((got[2][-1].cpu() - want[2][-1]).abs().max()):.2e}, translation {float((got[3][-1].cpu() - want[3][-1]).abs().max()):.2e} mm')
    m.decoder.iters = int(g['iters'])
    fr, fl, hf, cf = m.extract_feat(ginp['render_images'], ginp['real_images'])
    errs = {'feat_render': float((fr[:, ::8].cpu() - torch.from_numpy(g['feat_render'])).abs().max()),
            'h_feat': float((hf[:, ::8].cpu() - torch.from_numpy(g['h_feat'])).abs().max()),
            'cxt_feat': float((cf[:, ::8].cpu() - torch.from_numpy(g['cxt_feat'])).abs().max())}
    outs = m.get_pose(ginp['render_images'], ginp['real_images'], ginp['ref_rotation'], ginp['ref_translation'],
                      ginp['depth'], ginp['internel_k'], ginp['label'])
    for nm, seq in zip(names, outs):
        st = torch.stack(list(seq))
        if st.dim() == 5:
            st = st[..., ::4, ::4]
        errs[nm] = float((st.cpu() - torch.from_numpy(g[nm])).abs().max())
    print('   golden errors:', {k: f'{v:.2e}' for k, v in errs.items()})
m.decoder.iters = 8
for batch in (32, 8, 4, 1):
    b = bench.make_batch(batch, seed=1000, device=DEV)
    for wino in (False, True):
        ops.set_conv_winograd(wino)
        for _ in range(3):
            bench.run_step(m, b)
        torch.cuda.synchronize()
        reps = 10 if batch >= 8 else 30
        t0 = time.perf_counter()
        for _ in range(reps):
            bench.run_step(m, b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f'batch {batch} winograd={wino}: {dt * 1e3:.2f} ms/step, {batch / dt:.0f} pairs/s', flush=True)
