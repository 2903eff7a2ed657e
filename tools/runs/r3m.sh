#!/bin/bash
# default bench with the Winograd layers (2-D and F(2,5)) + kernel trace of a batch-32-only run
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p gpurun_out/r3m
python bench.py > gpurun_out/r3m/bench.json 2> gpurun_out/r3m/bench.err
tail -c 600 gpurun_out/r3m/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3m/trace -o t -- python $R/bench.py --no-alt --no-batch1 --no-config4 --no-cpu-baseline --steps 10 > $R/gpurun_out/r3m/trace.log 2>&1
cd $R
f=$(find gpurun_out/r3m/trace -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r3m/batch32_only_kernel_stats.csv
head -12 gpurun_out/r3m/batch32_only_kernel_stats.csv | cut -c1-160
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3m/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d.get('alt_direct'), d['alt_precision']['value'])
print(d['roofline'])
rc=d['roofline_conv']; print({k:v for k,v in rc.items() if k not in ('top_layers','note','gru_context_hoisting')})
for l in rc['top_layers']: print(l)
print(d.get('batch1')); print({k:v for k,v in d.get('config4',{}).items() if k in ('value','ms_per_step')})
PY
