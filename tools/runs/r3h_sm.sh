#!/bin/bash
# lookup store policy inside the real pipeline, v8 kernel: experiment builds of the library (tools/lab/build_exp.sh)
R=$GRAFT_REPO_ROOT; cd $R
cp scflow_amd/csrc/libscflow_hip.so /tmp/prod.so
for sm in 2 3 0 2 3; do
  cp tools/lab/bin/libscflow_hip_sm$sm.so scflow_amd/csrc/libscflow_hip.so
  timeout 200 python bench.py --steps 10 --warmup 3 --no-alt --no-batch1 --no-config4 --no-cpu-baseline --min-seconds 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('SM=$sm', 'pairs/s', d['value'], 'lookup avg us', r['avg_launch_us'], 'median', r['median_launch_us'], 'frac', r['frac'])"
done
cp /tmp/prod.so scflow_amd/csrc/libscflow_hip.so
