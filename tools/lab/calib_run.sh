#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE calibration passes (tools/lab/fetch_calib.hip), each bounded by a timeout
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-calib}
mkdir -p $OUT
C=$GRAFT_REPO_ROOT/tools/lab/bin/fetch_calib
timeout 120 $C > $OUT/plain_run.log 2>&1 || { echo "fetch_calib failed without the profiler"; tail -3 $OUT/plain_run.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o f -- $C > $OUT/calib_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o w -- $C > $OUT/calib_w.log 2>&1
find $OUT -name "*counter_collection.csv" | xargs ls -la
