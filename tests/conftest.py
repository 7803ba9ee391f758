import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# Engines place the submitting thread on the GPU's NUMA node only when asked (bind_host=True / NUHTC_HOST_AFFINITY=1) -- except the tools
# that own their process (tools/infer_wsi.py, tools/bench_wsi.py, bench.py), which ask unless NUHTC_HOST_AFFINITY=0.  The suite runs them
# as subprocesses beside the oracle's CPU work: keep them off.  The placement has its own tests (test_hip_edges.py, test_host.py).
os.environ.setdefault('NUHTC_HOST_AFFINITY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
