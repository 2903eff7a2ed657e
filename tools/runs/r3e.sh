#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
bash tools/collect_profiles.sh r3e > $O/collect.log 2>&1; tail -3 $O/collect.log
P=$R/gpurun_out/profiles_r3e
python tools/summarize_mfma.py $(find $P/pmc_mfma -name "*counter_collection.csv" | head -1) $(find $P/pmc_mfma -name "*kernel_trace.csv" | head -1) $O/mfma_pmc.json r3e 2>&1 | tail -30
python tools/summarize_pmc.py $(find $P/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $P/pmc_write -name "*counter_collection.csv" | head -1) $O/lookup_pmc.json r3e $(find $P/stats_f32 -name "*kernel_trace.csv" | head -1) > $O/summarize_pmc.log 2>&1; tail -5 $O/summarize_pmc.log
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3e/bench.json').read().strip().splitlines()[-1])
print('pairs/s', d['value'], 'ms', d['ms_per_step'], 'lookup', d['roofline']['avg_launch_us'], d['roofline']['frac'])
print('conv', d.get('roofline_conv',{}).get('achieved'), 'corr', d.get('roofline_corr_build'))
print('b1', d.get('batch1')); c4=d.get('config4',{}); print('c4', c4.get('value'), c4.get('roofline'), c4.get('roofline_corr_build'))
print('alt', d.get('alt_precision',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
