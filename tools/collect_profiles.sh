#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 evidence for bench.py, written under gpurun_out/.
# usage: tools/collect_profiles.sh <tag>
# Counter passes are their own runs (--pmc with --kernel-trace only, one counter per pass).
set -u
TAG=${1:-r2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --min-seconds 0.3 --min-warmup-seconds 0.1 --no-cpu-baseline --no-batch1 --no-alt"
# every profiler pass is time-bounded: a faulting child under rocprofv3 otherwise hangs until the box limit
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -o f32 -- $B --precision f32 > $OUT/bench_f32.log 2>&1
# (r6: no f16x3 pass any more -- the split-fp16 line is opt-in (bench.py --alt) and never a headline)
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $B > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- $B > /dev/null 2>&1
# matrix-core utilisation of the correlation GEMM and the convolution kernels (SQ: 7 of 8 slots, GRBM: 1 of 2)
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_mfma -o m -- $B --no-config4 > $OUT/pmc_mfma.log 2>&1
# calibration of both counters on the lookup's access types: tools/lab/calib_run.sh (own, time-bounded call)
rm -f $OUT/*/*_kernel_trace.csv.bak
find $OUT -name "*.csv" | head -40
