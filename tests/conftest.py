import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    config.addinivalue_line(
        'markers', 'gpu_experiments: a measured dead end that only libpbhip_exp.so carries '
        '(make -C pyratbay_amd/csrc EXPERIMENTS=1; PB_LIBPBHIP=pyratbay_amd/libpbhip_exp.so '
        'python -m pytest tests -m gpu_experiments); deselected with the default library')


def pytest_collection_modifyitems(config, items):
    """Tests of the experiments build are deselected unless that library is the one loaded."""
    import cases
    if cases.EXPERIMENTS:
        return
    keep, drop = [], []
    for item in items:
        (drop if item.get_closest_marker('gpu_experiments') else keep).append(item)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


@pytest.fixture(scope='session')
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle
