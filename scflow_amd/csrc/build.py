"""Build libscflow_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a
plain C-ABI shared object (include/scflow_hip.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['capi.hip', 'corr_lookup.hip', 'corr_gemm.hip', 'conv_mfma.hip', 'conv_f16x3.hip', 'conv_dma.hip', 'conv_thin.hip', 'conv_taps.hip', 'conv_wino.hip', 'conv_wino1d.hip', 'conv_wino1d4.hip', 'resample.hip', 'pose.hip', 'scflow_iter.hip',
           'norm.hip', 'metrics.hip', 'fc.hip']
OUT = os.path.join(HERE, 'libscflow_hip.so')
# conv_wino.hip: the SLP vectoriser turns the input transform's 32 adds into packed adds PLUS as many register
# moves to pair their operands up; vector-ALU instructions cost matrix-pipe time there (see the file), so the
# scalar form (no moves) is the faster one
FILE_FLAGS = {'conv_wino.hip': ['-fno-slp-vectorize'], 'conv_wino1d.hip': ['-fno-slp-vectorize'], 'conv_wino1d4.hip': ['-fno-slp-vectorize']}


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [
        os.path.join(HERE, 'scf_common.h'), os.path.join(HERE, 'conv_kernels.h'), os.path.join(HERE, 'scf_dma.h'), os.path.join(HERE, 'conv_taps_body.h'),
        os.path.join(HERE, '..', '..', 'include', 'scflow_hip.h'),
        os.path.join(HERE, '..', '..', 'include', 'scflow_hip_prof.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """one hipcc -c per source, in parallel (the MFMA convolution files dominate: ~45 s each
    alone), then one link."""
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
    if verbose:
        flags.append('-Rpass-analysis=kernel-resource-usage')
    objdir = os.path.join(HERE, '_obj')
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        subprocess.run([hipcc, *flags, *FILE_FLAGS.get(src, []), '-c', os.path.join(HERE, src), '-o', obj], check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', OUT], check=True)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
