#!/bin/bash
# GPU box: rocprofv3 kernel stats of the batch-1 pass (tools/bench_b1.py), time-bounded
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-b1prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b1 -- python $GRAFT_REPO_ROOT/tools/bench_b1.py > $OUT/b1.log 2>&1
tail -3 $OUT/b1.log | cut -c1-200
