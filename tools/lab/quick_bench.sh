#!/bin/bash
# quick headline check: conv-related GPU tests + one short bench line (summary on stdout)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv or gru" 2>&1 | tail -2
timeout 600 python bench.py --no-config4 --min-seconds 2 --top-layers ${TOPL:-14} > gpurun_out/quick.json 2> gpurun_out/quick.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'conv', d['roofline_conv']['achieved'], 'b1', d['batch1']['ms_per_pair_hipgraph'], d['batch1']['ms_per_pair_eager'])
for l in d['roofline_conv']['top_layers']: print(f"  {l['layer']:42s} x{l['launches']:3d} {l['us']:8.1f}us {l['tflops']:6.1f}TF")
PY
