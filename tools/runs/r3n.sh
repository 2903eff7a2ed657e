#!/bin/bash
# end-of-round evidence: GPU tests, smoke, default bench, rocprofv3 passes (stats f32 / f16x3, FETCH, WRITE, SQ MFMA)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
bash tools/collect_profiles.sh r3n > $O/collect.log 2>&1; tail -2 $O/collect.log
P=$R/gpurun_out/profiles_r3n
python tools/summarize_mfma.py $(find $P/pmc_mfma -name "*counter_collection.csv" | head -1) $(find $P/pmc_mfma -name "*kernel_trace.csv" | head -1) $O/mfma_pmc.json r3n 2>&1 | tail -16
python tools/summarize_pmc.py $(find $P/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $P/pmc_write -name "*counter_collection.csv" | head -1) $O/lookup_pmc.json r3n $(find $P/stats_f32 -name "*kernel_trace.csv" | head -1) > $O/summarize_pmc.log 2>&1; grep -n "traffic_bytes\|avg_us" $O/summarize_pmc.log
bash tools/lab/b32_profile.sh r3n_b32 > /dev/null 2>&1
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3n/bench.json').read().strip().splitlines()[-1])
print('pairs/s', d['value'], 'ms', d['ms_per_step'], 'lookup', d['roofline']['avg_launch_us'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
rc=d['roofline_conv']; print('conv', rc['achieved'], rc['frac'], rc['conv_us_per_step'], rc['gru_context_hoisting']['tflops_in_reference_formulation'])
print('corr', d['roofline_corr_build']['achieved'], d['roofline_corr_build']['frac'], d['roofline_corr_build']['standalone_level0_only']['frac'])
print('b1', d.get('batch1')); c4=d.get('config4',{}); print('c4', c4.get('value'), c4.get('ms_per_step'), c4.get('roofline',{}).get('avg_launch_us'), c4.get('roofline',{}).get('frac'), c4['roofline_corr_build']['frac'])
print('alt', d.get('alt_precision',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
