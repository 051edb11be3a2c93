import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the GPU tests replay hipGraphs in this process (rollout graph, asynchronous learner): the package refuses to do that unless
# the HIP runtime's graph packet capture was switched off before the first HIP call (shogidrl_amd/graphs.py) -- i.e. here
from shogidrl_amd import graphs  # noqa: E402

graphs.enable_safe_replay()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
