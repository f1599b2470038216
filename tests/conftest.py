import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
GOLDEN_CASES = ['n6_p1', 'n5_p4', 'n5_p2_ecstr', 'n4_p6_pbc', 'n9_p1', 'n10_p2_pbc']


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    import numpy as np

    g = dict(np.load(os.path.join(GOLDEN_DIR, request.param + '.npz')))
    g['name'] = request.param
    return g


def pytest_runtest_logstart(nodeid, location):
    """Breadcrumb for post-mortems of hard crashes (GPU memory faults abort the interpreter)."""
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        try:
            with open(os.path.join(d, 'last_test.txt'), 'w') as f:
                f.write(nodeid + '\n')
        except OSError:
            pass


@pytest.fixture(scope='session', autouse=True)
def _gpu_first_touch():
    """GPU sessions: a child process takes the first touch of the device (sgdml_amd._lib.preflight: the first kernel launch
    on a freshly leased box has aborted inside the HIP runtime a few times; retried there, not in the test process)."""
    try:
        from sgdml_amd import _lib

        if _lib.device_count() > 0:
            _lib.preflight()
    except Exception:  # no library / no GPU / preflight exhausted: let the tests speak
        pass
    yield
