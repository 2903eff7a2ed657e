import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scflow_amd
from scflow_amd import ops
import scflow_amd.refiner as R
from scflow_amd.graph import GraphedRefiner
DEV = 'cuda:0'
shapes = json.load(open('tests/golden/state_dict_keys.json'))['shapes']
m = scflow_amd.build_refiner(scflow_amd.scflow_model_cfg(iters=3))
m.load_state_dict(scflow_amd.fill_state_dict(shapes, seed=0), strict=True)
m = m.to(DEV)
a = {k: v.to(DEV) for k, v in scflow_amd.make_inputs(2, 256, 256, seed=21).items()}
names = ['flow_from_pose', 'flow_from_pred', 'rot', 'trans', 'mask', 'd_rot', 'd_trans']
def runi(i):
    return m.get_pose(i['render_images'], i['real_images'], i['ref_rotation'], i['ref_translation'], i['depth'], i['internel_k'], i['label'])
def report(x, y, tag):
    out = []
    for nm, sx, sy in zip(names, x, y):
        for i, (tx, ty) in enumerate(zip(sx, sy)):
            if not torch.equal(tx, ty):
                out.append(f'{nm}[{i}] {float((tx - ty).abs().max()):.3g}')
    print(tag, 'identical' if not out else ' '.join(out))
real_small = ops.small_work
for mode in ('both', 'refiner only', 'decoder only', 'none'):
    R.small_work = real_small if mode in ('both', 'refiner only') else (lambda *a_: False)
    ops.small_work = real_small if mode in ('both', 'decoder only') else (lambda *a_: False)
    want = [[t.clone() for t in s_] for s_ in runi(a)]
    g = GraphedRefiner(m, a)
    got = g(a)
    torch.cuda.synchronize()
    report(want, got, f'[{mode}] eager vs graph:')
    del g
