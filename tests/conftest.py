import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def tiny():
    return load_golden('tiny_model.npz')


@pytest.fixture(scope='session')
def host_helpers():
    return load_golden('host_helpers.npz')


@pytest.fixture(scope='session')
def trainer_steps():
    return load_golden('trainer_steps.npz')


@pytest.fixture(scope='session')
def shapes_base():
    return load_golden('shapes_base.npz')


@pytest.fixture(scope='session')
def shapes_large():
    return load_golden('shapes_large.npz')
