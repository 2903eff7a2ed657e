"""graph replay vs eager, many times, with / without side-stream overlap (experiment)."""
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
ins = [{k: v.to(DEV) for k, v in scflow_amd.make_inputs(2, 256, 256, seed=21 + i).items()} for i in range(3)]
def runi(i):
    return m.get_pose(i['render_images'], i['real_images'], i['ref_rotation'], i['ref_translation'], i['depth'], i['internel_k'], i['label'])
for mode in (set(), {'context'}, {'flow'}, {'mask'}, {'upsample'}, {'context', 'flow', 'mask', 'upsample'}):
    ops.OVERLAP_BRANCHES.clear(); ops.OVERLAP_BRANCHES.update(mode)
    wants = [[[t.clone() for t in s_] for s_ in runi(i)] for i in ins]
    bad_e = 0
    for rep in range(30):
        j = rep % 3
        e = runi(ins[j])
        torch.cuda.synchronize()
        bad_e += not all(torch.equal(a, b) for sa, sb in zip(e, wants[j]) for a, b in zip(sa, sb))
    g = GraphedRefiner(m, ins[0])
    bad_g = 0
    for rep in range(30):
        j = rep % 3
        got = g(ins[j])
        torch.cuda.synchronize()
        bad_g += not all(torch.equal(a, b) for sa, sb in zip(got, wants[j]) for a, b in zip(sa, sb))
    print(f'overlap {sorted(mode)}: wrong eager {bad_e}/30, wrong graph replays {bad_g}/30')
    del g
