#!/bin/bash
# experiment build of the library: tools/lab/build_exp.sh "<extra -D flags>" [suffix] -> tools/lab/bin/libscflow_hip_exp[suffix].so
set -e
cd "$(dirname "$0")/../../scflow_amd/csrc"
O=/tmp/scf_exp_obj$2; mkdir -p $O
for f in capi corr_lookup corr_gemm conv_mfma conv_f16x3 conv_dma conv_thin conv_taps conv_wino conv_wino1d resample pose norm metrics scflow_iter; do
  X=""; [ $f = conv_wino -o $f = conv_wino1d ] && X="-fno-slp-vectorize"      # as scflow_amd/csrc/build.py FILE_FLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../tools/lab $X $1 -c $f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -o ../../tools/lab/bin/libscflow_hip_exp$2.so
ls -la ../../tools/lab/bin/libscflow_hip_exp$2.so
