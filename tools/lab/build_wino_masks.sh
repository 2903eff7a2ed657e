#!/bin/bash
# compile-time ablations of the Winograd kernel: tools/lab/build_wino_masks.sh "0 1 2 4 6 7 16 22" ->
# tools/lab/bin/libscflow_hip_exp_m<mask>.so (only conv_wino.hip is recompiled per mask)
set -e
cd "$(dirname "$0")/../../scflow_amd/csrc"
O=/tmp/scf_exp_obj_base; mkdir -p $O
for f in capi corr_lookup corr_gemm conv_mfma conv_f16x3 conv_dma conv_thin conv_taps conv_wino1d resample pose norm metrics scflow_iter; do
  [ $O/$f.o -nt $f.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../tools/lab -c $f.hip -o $O/$f.o &
done
wait
for m in $1; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../tools/lab -fno-slp-vectorize -DSCF_WINO_LAB -DSCF_WINO_LAB_MASK=$m -c conv_wino.hip -o /tmp/scf_wino_m$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o /tmp/scf_wino_m$m.o -o ../../tools/lab/bin/libscflow_hip_exp_m$m.so ) &
done
wait
ls ../../tools/lab/bin/
