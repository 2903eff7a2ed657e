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
