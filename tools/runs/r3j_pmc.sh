#!/bin/bash
# SQ counters of the Winograd kernel (one layer shape), two passes
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p gpurun_out/wino_pmc
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/wino_pmc/a -o a -- python $R/tools/lab/wino_layers.py "128->512 @32 N32" > $R/gpurun_out/wino_pmc/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/wino_pmc/b -o b -- python $R/tools/lab/wino_layers.py "128->512 @32 N32" > $R/gpurun_out/wino_pmc/b.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for tag in 'ab':
    for f in glob.glob(f'gpurun_out/wino_pmc/{tag}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            if 'conv_' not in k: continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in agg.items():
            print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, 'dispatches', len(next(iter(d.values()))))
PY
tail -3 gpurun_out/wino_pmc/a.log gpurun_out/wino_pmc/b.log
