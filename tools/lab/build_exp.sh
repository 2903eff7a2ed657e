#!/bin/bash
# experiment build of the library: tools/lab/build_exp.sh "<extra -D flags>" -> tools/lab/bin/libscflow_hip_exp.so
set -e
cd "$(dirname "$0")/../../scflow_amd/csrc"
O=/tmp/scf_exp_obj; mkdir -p $O
for f in capi corr_lookup corr_gemm conv_mfma conv_f16x3 conv_dma conv_thin conv_taps conv_wino resample pose norm scflow_iter; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $1 -c $f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -o ../../tools/lab/bin/libscflow_hip_exp.so
ls -la ../../tools/lab/bin/libscflow_hip_exp.so
