#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 900 python bench.py --top-layers 40 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3g/bench.json').read().strip().splitlines()[-1])
print('pairs/s', d['value'], 'ms', d['ms_per_step'], 'lookup', d['roofline']['avg_launch_us'], d['roofline']['frac'])
rc=d['roofline_conv']; print('conv', rc['achieved'], rc['conv_us_per_step'], 'b1', d.get('batch1'))
c4=d.get('config4',{}); print('c4', c4.get('value'), c4.get('roofline',{}).get('frac')); print('alt', d.get('alt_precision',{}).get('value'))
for l in rc['top_layers']:
    if any(t in l['layer'] for t in ('224->128','128->128 3x3/s2', '64->32')): print('  ', l)
PY
for n in 1 2 4 8 16; do python tools/bench_b1.py $n 2>&1 | grep -v amdgpu; done
