import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scflow_amd
from scflow_amd import ops
from scflow_amd.graph import GraphedRefiner
DEV = 'cuda:0'
shapes = json.load(open('tests/golden/state_dict_keys.json'))['shapes']
m = scflow_amd.build_refiner(scflow_amd.scflow_model_cfg(iters=3))
m.load_state_dict(scflow_amd.fill_state_dict(shapes, seed=0), strict=True)
m = m.to(DEV)
a = {k: v.to(DEV) for k, v in scflow_amd.make_inputs(2, 256, 256, seed=21).items()}
run = lambda: m.get_pose(a['render_images'], a['real_images'], a['ref_rotation'], a['ref_translation'], a['depth'], a['internel_k'], a['label'])
names = ['flow_from_pose', 'flow_from_pred', 'rot', 'trans', 'mask', 'd_rot', 'd_trans']
def cmp(x, y, tag):
    out = []
    for nm, sx, sy in zip(names, x, y):
        for i, (tx, ty) in enumerate(zip(sx, sy)):
            if not torch.equal(tx, ty):
                out.append(f'{nm}[{i}] {float((tx - ty).abs().max()):.3g}')
    print(tag, 'identical' if not out else ' '.join(out))
    return not out
r1 = [[t.clone() for t in s] for s in run()]
r2 = [[t.clone() for t in s] for s in run()]
cmp(r1, r2, 'eager vs eager:')
b = {k: v.to(DEV) for k, v in scflow_amd.make_inputs(2, 256, 256, seed=22).items()}
def runi(i):
    return m.get_pose(i['render_images'], i['real_images'], i['ref_rotation'], i['ref_translation'], i['depth'], i['internel_k'], i['label'])
g = GraphedRefiner(m, a)
for tag, inp in (('a', a), ('b', b), ('a2', a)):
    want = [[t.clone() for t in s_] for s_ in runi(inp)]
    got = g(inp)
    torch.cuda.synchronize()
    cmp(want, got, f'overlap ON, eager vs graph [{tag}]:')
    if tag != 'b':
        cmp(r1, want, f'   eager now vs eager before the graph existed [{tag}]:')
        cmp(r1, got, f'   graph vs eager before the graph existed [{tag}]:')
    got2 = [[t.clone() for t in s_] for s_ in g(inp)]
    torch.cuda.synchronize()
    got3 = g(inp)
    torch.cuda.synchronize()
    cmp(got2, got3, f'overlap ON, graph vs graph [{tag}]:')
