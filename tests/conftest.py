import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """Multi-process GPU tests first: their children must be started while this process has not yet
    initialised the GPU (a GPU-initialised parent must not fork+exec on the GPU pool)."""
    items.sort(key=lambda it: 0 if 'test_gpu_multiproc' in it.nodeid else 1)      # stable


@pytest.fixture(scope='session')
def require_gpu():
    """GPU tests must fail loudly (not skip) when the HIP path cannot run."""
    from monte_carlo_gp_amd import _native
    n = _native.lib().mcgp_device_count()
    assert n > 0, 'no HIP device visible: the -m gpu tests need a GPU'
    return n
