import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope='session')
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _persistent_kernels_completed(request):
    """After every GPU test: fail it if a persistent recurrent kernel gave up waiting for a peer workgroup."""
    yield
    if request.node.get_closest_marker('gpu') is not None:
        from morgana_amd import ops
        ops.check_persistent_status()


def pytest_sessionfinish(session, exitstatus):
    """The GPU suite's observed parity errors -> gpurun_out/parity_report.json (tests/parity_report.py)."""
    import parity_report
    if parity_report.RECORDS:
        try:
            parity_report.write(os.path.join(REPO, 'gpurun_out', 'parity_report.json'))
        except OSError:
            pass
