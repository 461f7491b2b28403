import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """Tests marked gpu need a HIP device: without one they are skipped (not failed), so that a plain `pytest tests` runs the
    whole CPU suite instead of stopping at the first GPU test."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason='no HIP device in this container (run on the GPU box with -m gpu)')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle.lib()
