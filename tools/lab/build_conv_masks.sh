#!/bin/bash
# compile-time ablations of the direct LDS-DMA convolution kernel: tools/lab/build_conv_masks.sh "0 1 2 4 5" ->
# tools/lab/bin/libscflow_hip_exp_c<mask>.so (only conv_dma.hip is recompiled per mask)
set -e
cd "$(dirname "$0")/../../scflow_amd/csrc"
O=/tmp/scf_exp_obj_cbase; mkdir -p $O
for f in capi corr_lookup corr_gemm conv_mfma conv_f16x3 conv_wino conv_wino1d conv_thin conv_taps resample pose norm metrics scflow_iter; do
  X=""; [ $f = conv_wino -o $f = conv_wino1d ] && X="-fno-slp-vectorize"
  [ $O/$f.o -nt $f.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../tools/lab $X -c $f.hip -o $O/$f.o &
done
wait
for m in $1; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../tools/lab -DSCF_CONV_LAB -DSCF_CONV_LAB_MASK=$m -c conv_dma.hip -o /tmp/scf_conv_c$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o /tmp/scf_conv_c$m.o -o ../../tools/lab/bin/libscflow_hip_exp_c$m.so ) &
done
wait
