#!/bin/bash
# round 3, first GPU pass: full GPU test suite with measured errors, sector-granularity timing of the
# calibration patterns, correlation micro-benchmark in both layouts, one default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
grep "measured" $O/pytest.log | sort | uniq | head -80
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/calib -o c -- $R/tools/lab/bin/fetch_calib > $O/calib.log 2>&1
find $O/calib -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
cd $R
timeout 600 python tools/microbench_corr.py > $O/microbench_corr.txt 2>&1; cat $O/microbench_corr.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3a/bench.json').read().strip().splitlines()[-1])
print('pairs/s', d['value'], 'ms', d['ms_per_step'], 'lookup', d['roofline']['avg_launch_us'], d['roofline']['frac'])
print('conv', d.get('roofline_conv',{}).get('achieved'), 'corr', d.get('roofline_corr_build',{}).get('achieved'))
print('b1', d.get('batch1')); c4=d.get('config4',{}); print('c4', c4.get('value'), c4.get('roofline'))
print('cpu', d.get('cpu_baseline'))
PY
