#!/bin/bash
# GPU box: rocprofv3 kernel stats of the batch-32 step alone (no configs[4] / batch-1 / split-fp16 legs), time-bounded
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-b32prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b32 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --min-seconds 0.2 --min-warmup-seconds 0.1 --no-cpu-baseline --no-batch1 --no-alt --no-config4 > $OUT/b32.log 2>&1
tail -2 $OUT/b32.log | cut -c1-200
