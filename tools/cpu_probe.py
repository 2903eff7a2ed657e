import os, time, json, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oracle, scflow_amd
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
try:
    print(open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e:
    print('no cgroup cpu.max', e)
shapes = json.load(open('tests/golden/state_dict_keys.json'))['shapes']
sd = scflow_amd.fill_state_dict(shapes, seed=0)
inp = scflow_amd.make_inputs(1, 256, 256, seed=99)
args = (inp['render_images'], inp['real_images'], inp['ref_rotation'], inp['ref_translation'], inp['depth'], inp['internel_k'], inp['label'], sd)
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    with torch.no_grad():
        oracle.get_pose(*args, iters=1)
        t0=time.perf_counter(); oracle.get_pose(*args, iters=8); dt=time.perf_counter()-t0
    print(th, 'threads:', round(dt,3), 's/pair', flush=True)
    if dt > 20: break
