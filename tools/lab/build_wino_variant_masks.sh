#!/bin/bash
# compile-time ablations of the F(2x2,3x3) kernels per variant: tools/lab/build_wino_variant_masks.sh "2 3" "0 1 2 4 6 7 32" ->
# tools/lab/bin/libscflow_hip_exp_v<variant>m<mask>.so (only conv_wino.hip is recompiled per build)
set -e
cd "$(dirname "$0")/../../scflow_amd/csrc"
mkdir -p ../../tools/lab/bin
O=_obj
for v in $1; do for m in $2; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../tools/lab -fno-slp-vectorize -DSCF_WINO_LAB -DSCF_WINO_LAB_MASK=$m -DSCF_WINO_DEFAULT_VARIANT=$v -c conv_wino.hip -o /tmp/scf_wino_v${v}m$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $O/*.o | grep -v conv_wino.o) /tmp/scf_wino_v${v}m$m.o -o ../../tools/lab/bin/libscflow_hip_exp_v${v}m$m.so ) &
done; done
wait
ls ../../tools/lab/bin/ | grep _v
