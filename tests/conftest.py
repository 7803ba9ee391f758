import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# Engine() moves the calling thread onto the GPU's NUMA node (hip.bind_host_thread); the suite's CPU work (the oracle) would then run on half
# of the host for the rest of the session.  The placement has its own test (test_hip_edges.py), which switches it on.
os.environ.setdefault('NUHTC_HOST_AFFINITY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
