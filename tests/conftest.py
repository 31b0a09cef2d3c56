import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


os.environ["KSOLVE_TEST_SOLVER_LIB"] = "1"   # tests may hand a test build of the solver library to NewScheduler(solver_lib=) (karpenter_amd/scheduling.py gates it)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as _oracle  # test infrastructure only

    _oracle.build()
    return _oracle
