#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 900 python bench.py --top-layers 40 --no-config4 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3f/bench.json').read().strip().splitlines()[-1])
print('pairs/s', d['value'], 'ms', d['ms_per_step'], 'lookup', d['roofline']['avg_launch_us'], d['roofline']['frac'])
rc=d['roofline_conv']; print('conv', rc['achieved'], rc['conv_us_per_step'], 'b1', d.get('batch1'))
for l in rc['top_layers']:
    if any(t in l['layer'] for t in ('7x7','1->64','224->128','128->128 3x3/s2')): print('  ', l)
PY
