#!/bin/bash
# GPU box: rocprofv3 kernel stats of the configs[4] step alone (bench.config4_block, 0.5 s), time-bounded
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-c4prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c4 -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import bench, json
r = bench.config4_block('cuda:0', 0.5)
print(json.dumps(r)[:300])
" > $OUT/c4.log 2>&1
tail -2 $OUT/c4.log | cut -c1-300
