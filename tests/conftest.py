import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gold_dir():
    return GOLD


@pytest.fixture(scope='session')
def sncal():
    """The product package (fails loudly if libsncal.so is missing)."""
    import sncal_amd
    sncal_amd._lib.lib()
    return sncal_amd


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('gpu-marked test needs a visible GPU')
    return torch.device('cuda:0')
