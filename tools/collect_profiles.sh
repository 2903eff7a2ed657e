#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 evidence for bench.py, written under gpurun_out/.
# usage: tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-alt"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -o f32 -- $B --precision f32 > $OUT/bench_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f16x3 -o f16x3 -- $B --precision f16x3 > $OUT/bench_f16x3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- $B > /dev/null 2>&1
rm -f $OUT/*/*_kernel_trace.csv.bak
ls -R $OUT | head -30
