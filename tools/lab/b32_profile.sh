#!/bin/bash
# GPU box: rocprofv3 kernel stats of STEADY-STATE batch-32 steps (tools/steady_trace.py runs steps back to back; the
# collection window 40 s .. 42 s after start holds no warm-up, packing or timer set-up launch), time-bounded
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-b32prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --collection-period 40:2:1 --output-format csv -d $OUT -o b32 -- python $GRAFT_REPO_ROOT/tools/steady_trace.py ${2:-32} 44 > $OUT/b32.log 2>&1
tail -2 $OUT/b32.log | cut -c1-200
