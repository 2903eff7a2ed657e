#!/bin/bash
# GPU box: steady-state kernel trace of hipGraph replays of the batch-N pass + tools/lab/b1_timeline.py on it
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-b1tl}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GRAPH=1 timeout 600 rocprofv3 --kernel-trace --collection-period 40:1:1 --output-format csv -d /tmp/b1tl -o b1 -- python $GRAFT_REPO_ROOT/tools/steady_trace.py ${2:-1} 43 > $OUT/trace.log 2>&1
F=$(find /tmp/b1tl -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/lab/b1_timeline.py $F ${3:-400} > $OUT/timeline.txt 2>&1
tail -40 $OUT/timeline.txt
