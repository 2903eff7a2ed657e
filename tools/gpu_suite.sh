#!/bin/bash
# Run ON THE GPU BOX (via gpurun): tools/gpu_suite.sh <tag> <stage> [<stage> ...]; everything lands in gpurun_out/<tag>/.
#   tests      the whole -m gpu suite (checks the pinned dispatch plan)
#   plan       the -m gpu suite with SCF_WRITE_DISPATCH_PLAN=1 (records gpurun_out/dispatch_plan.json), then re-checks
#              the parity tests that carry a plan against the freshly recorded one
#   measured   the tests that print [measured] lines (Winograd stress, GRU drift, configs[2]/[4]) with -s
#   smoke      __graft_entry__.smoke()
#   bench      python bench.py (defaults)
#   profiles   tools/collect_profiles.sh <tag>, the counter summaries, the batch-32-only kernel trace, tools/lab/coissue
#              (tools/publish_profiles.py <tag> then copies what is cited into profiles/)
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for stage in "$@"; do
  case $stage in
    tests)
      timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $OUT/tests.log; tail -5 $OUT/tests.log ;;
    plan)
      rm -f $R/gpurun_out/dispatch_plan.json
      SCF_WRITE_DISPATCH_PLAN=1 timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $OUT/tests_plan.log; tail -15 $OUT/tests_plan.log
      # the recorded plan is NOT copied over the pinned one: it is diffed here and re-checked from gpurun_out/; a reviewed
      # plan is committed by hand (cp gpurun_out/dispatch_plan.json tests/golden/)
      python - <<'PY' | tee $OUT/plan_diff.log
import json
a = json.load(open('tests/golden/dispatch_plan.json')); b = json.load(open('gpurun_out/dispatch_plan.json'))
for k in sorted(set(a) | set(b)):
    if a.get(k) != b.get(k):
        print('PLAN DIFF', k, 'pinned-only:', sorted(set(a.get(k) or []) - set(b.get(k) or [])) if isinstance(a.get(k), list) else a.get(k),
              'recorded-only:', sorted(set(b.get(k) or []) - set(a.get(k) or [])) if isinstance(b.get(k), list) else b.get(k))
print('plan diff done')
PY
      SCF_DISPATCH_PLAN=$R/gpurun_out/dispatch_plan.json timeout 1200 python -m pytest tests/test_gpu_refiner.py tests/test_next_rows.py -m gpu -q -k "golden or config2 or config4" 2>&1 | tail -5 > $OUT/tests_plan_check.log; tail -3 $OUT/tests_plan_check.log ;;
    measured)
      timeout 1800 python -m pytest tests -m gpu -q -s -k "stress or drift or config2 or config4_full or winograd" 2>&1 | grep -a "measured\|passed\|failed\|Error\|assert" > $OUT/measured.log; tail -60 $OUT/measured.log ;;
    smoke)
      timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.log ;;
    bench)
      timeout 900 python bench.py --top-layers 80 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    profiles)      # rocprofv3 passes of bench.py (stats f32 / f16x3, FETCH, WRITE, SQ MFMA), summaries, batch-32-only trace, co-issue lab
      bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1; tail -2 $OUT/collect.log
      P=$R/gpurun_out/profiles_$TAG
      python tools/summarize_mfma.py $(find $P/pmc_mfma -name "*counter_collection.csv" | head -1) $(find $P/pmc_mfma -name "*kernel_trace.csv" | head -1) $OUT/mfma_pmc.json $TAG 2>&1 | tail -14
      python tools/summarize_pmc.py $(find $P/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $P/pmc_write -name "*counter_collection.csv" | head -1) $OUT/lookup_pmc.json $TAG $(find $P/stats_f32 -name "*kernel_trace.csv" | head -1) > $OUT/summarize_pmc.log 2>&1; grep -n "traffic_bytes\|avg_us" $OUT/summarize_pmc.log
      bash tools/lab/b32_profile.sh ${TAG}_b32 > /dev/null 2>&1
      [ -x tools/lab/bin/coissue ] && timeout 120 tools/lab/bin/coissue > $OUT/coissue.txt 2>&1 ;;
    *) echo "unknown stage $stage" ;;
  esac
done
