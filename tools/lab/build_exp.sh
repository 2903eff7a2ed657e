#!/bin/bash
# experiment build of the library: tools/lab/build_exp.sh "<extra -D flags>" [suffix] -> tools/lab/bin/libscflow_hip_exp[suffix].so
# (sources and per-file flags as scflow_amd/csrc/build.py has them)
set -e
cd "$(dirname "$0")/../../scflow_amd/csrc"
O=/tmp/scf_exp_obj$2; mkdir -p $O ../../tools/lab/bin
for f in $(python3 -c "import build; print(' '.join(s[:-4] for s in build.SOURCES))"); do
  X=$(python3 -c "import build; print(' '.join(build.FILE_FLAGS.get('$f.hip', [])))")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../tools/lab $X $1 -c $f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -o ../../tools/lab/bin/libscflow_hip_exp$2.so
ls -la ../../tools/lab/bin/libscflow_hip_exp$2.so
