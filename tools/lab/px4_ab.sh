#!/bin/bash
# A/B of the aligned-x4 patch staging: product build (PX4 on) vs tools/lab/bin/libscflow_hip_exp.so (built with -DSCF_PX4_MODE=0)
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv or gru" 2>&1 | tail -5
timeout 600 python bench.py --no-config4 --min-seconds 2 > gpurun_out/px4_on.json 2> gpurun_out/px4_on.err
cp scflow_amd/csrc/libscflow_hip.so /tmp/prod.so
cp tools/lab/bin/libscflow_hip_exp.so scflow_amd/csrc/libscflow_hip.so
timeout 600 python bench.py --no-config4 --min-seconds 2 > gpurun_out/px4_off.json 2> gpurun_out/px4_off.err
cp /tmp/prod.so scflow_amd/csrc/libscflow_hip.so
python - <<'PY'
import json
for t in ('on','off'):
    try:
        d=json.loads(open(f'gpurun_out/px4_{t}.json').read().strip().splitlines()[-1])
        print(t, d['value'], d['ms_per_step'], 'conv', d.get('roofline_conv',{}).get('achieved'), 'b1', d.get('batch1',{}))
    except Exception as e:
        print(t, 'ERR', e)
PY
