#!/bin/bash
# tools/lab/lib_ab.sh <lib.so> ...: one short bench line per library build (first: the product build)
cp scflow_amd/csrc/libscflow_hip.so /tmp/prod.so
for L in prod "$@"; do
  [ "$L" != prod ] && cp $L scflow_amd/csrc/libscflow_hip.so
  timeout 600 python bench.py --no-config4 --no-alt --no-cpu-baseline --min-seconds 2 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$L" <<'PY'
import json, sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], 'conv', d['roofline_conv']['achieved'], 'b1', d['batch1']['ms_per_pair_hipgraph'])
PY
done
cp /tmp/prod.so scflow_amd/csrc/libscflow_hip.so
