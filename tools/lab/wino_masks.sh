#!/bin/bash
# run tools/lab/wino_phases.py on every compile-time ablation build
cd "$(dirname "$0")/../.."
for m in ${MASKS:-0 1 2 4 6 7 16 22 23}; do
  SCF_EXP_SUFFIX=_m$m timeout 120 python tools/lab/wino_phases.py 2>&1 | grep -A1 "64->64\|128->512" | grep -v "^--" | paste - - - -
done
