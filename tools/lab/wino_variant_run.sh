#!/bin/bash
# GPU box: correctness + timings of the F(2x2,3x3) variants, then the phase ablations of each (lab builds)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-wv}; mkdir -p $OUT
timeout 600 python tools/lab/wino_variants.py check > $OUT/check.log 2>&1; tail -20 $OUT/check.log
timeout 900 python tools/lab/wino_variants.py time > $OUT/time.log 2>&1; cat $OUT/time.log
for v in 1 2 3; do for m in 0 1 2 32 7; do
  echo "== variant $v mask $m" >> $OUT/masks.log
  SCF_EXP_SUFFIX=_v${v}m$m SCF_VARIANTS=0 timeout 300 python tools/lab/wino_variants.py time "128->512 @32 N32" "64->64 @128 N64" "256->192 @32 N32" >> $OUT/masks.log 2>&1
done; done
cat $OUT/masks.log
