#!/bin/bash
# TEMPORARY experiment driver (GPU box): lookup store policy inside the real pipeline
cp tools/lab/bin/libscflow_hip_exp.so scflow_amd/csrc/libscflow_hip.so
for sm in 2 0 1 4 3 2; do
  SCF_LK_SM=$sm timeout 200 python bench.py --steps 10 --warmup 3 --no-alt --no-batch1 --no-config4 --no-cpu-baseline --min-seconds 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('SM=$sm', 'pairs/s', d['value'], 'lookup avg us', r['avg_launch_us'], 'median', r['median_launch_us'], 'frac', r['frac'])"
done
