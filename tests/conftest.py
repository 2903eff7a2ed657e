import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _library_built():
    """Build libscflow_hip.so once if it is missing or stale and hipcc is available (the .so is
    git-ignored; normally ``__graft_entry__.build()`` has produced it already)."""
    import importlib.util
    import shutil
    path = os.path.join(ROOT, 'scflow_amd', 'csrc', 'build.py')
    spec = importlib.util.spec_from_file_location('_scf_build', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if mod.needs_build() and (os.path.exists(hipcc) or shutil.which('hipcc')):
        mod.build()
    yield


# ---- which convolution kernel ran --------------------------------------------------
# Kernel selection (Winograd vs direct, pixel-split vs K-split tiles) depends on the grid size and the CU count, so
# a moved threshold could silently put a golden test on other arithmetic.  Parity tests record the launches of their
# HIP run (ops.record_conv_kernels: the library's own dispatch log) and compare the SET of (layer, kernel family)
# with the plan pinned in tests/golden/dispatch_plan.json (MI355X, 256 CUs).  SCF_WRITE_DISPATCH_PLAN=1 (on the GPU
# box) records instead of checking and writes gpurun_out/dispatch_plan.json, which is then reviewed and committed.
DISPATCH_PLAN = os.path.join(GOLDEN, 'dispatch_plan.json')


def _cu_count():
    import torch
    return int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)


def check_dispatch(name, ran):
    """the plan is a property of (library, device): it carries the CU count it was recorded on ('_cus'), and a
    device with another CU count (a partitioned MI355X, another gfx950 SKU) skips the comparison instead of failing
    every golden test on it -- the numerics assertions of the calling test still run."""
    import json
    got = sorted({f'{tag} | {kind}' for tag, kind in ran})
    assert got, f'{name}: no convolution launch was recorded'
    if os.environ.get('SCF_WRITE_DISPATCH_PLAN'):
        out = os.path.join(ROOT, 'gpurun_out', 'dispatch_plan.json')
        os.makedirs(os.path.dirname(out), exist_ok=True)
        plan = json.load(open(out)) if os.path.exists(out) else {}
        plan['_cus'] = _cu_count()
        plan[name] = got
        with open(out, 'w') as f:
            json.dump(plan, f, indent=1, sort_keys=True)
        return
    plan = json.load(open(os.environ.get('SCF_DISPATCH_PLAN', DISPATCH_PLAN)))
    if plan.get('_cus', 256) != _cu_count():
        import warnings
        warnings.warn(f"{name}: dispatch plan recorded on {plan.get('_cus', 256)} CUs, this device has {_cu_count()}: "
                      'kernel-selection check skipped')
        return
    assert name in plan, f'{name}: no pinned dispatch plan (run the GPU tests with SCF_WRITE_DISPATCH_PLAN=1)'
    want = plan[name]
    assert got == want, (f'{name}: the convolution kernels that ran differ from the pinned plan -- '
                         f'only in this run: {sorted(set(got) - set(want))}; only in the plan: {sorted(set(want) - set(got))}')


@pytest.fixture(scope='session')
def dispatch_check():
    return check_dispatch
