#!/bin/bash
# Run ON THE GPU BOX: SQ counters for the conv micro-benchmark layers (fp32 path only).
# usage: tools/pmc_conv.sh <tag> [layer-name-substring ...]
set -u
TAG=${1:-conv}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/microbench_conv.py $*"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for tag in 'ab':
    fs = glob.glob('$OUT/%s/**/*counter_collection.csv' % tag, recursive=True)
    if not fs:
        print('no counter file for pass', tag); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if 'conv_' not in k: continue
        key = (k.split('(')[0], r.get('Grid_Size'), r.get('LDS_Block_Size'))
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key, d in acc.items():
        print(key)
        for c, v in sorted(d.items()):
            v = sorted(v); print('   %-28s median %14.0f  (n=%d)' % (c, v[len(v)//2], len(v)))
PY
