import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'image-restoration-agent_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _bounded_cpu_threads():
    """The CPU oracles are torch-CPU code.  torch defaults to one thread per visible core, but a one-GPU job on the GPU box
    has a 16-core share of a much larger host: without a bound the OpenMP workers oversubscribe and a 20 s oracle call
    takes minutes (seen as a 'hang' of the suite)."""
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    yield


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
