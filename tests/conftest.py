import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=True)


@pytest.fixture(scope="session")
def golden_ops():
    return load_golden("ops")


@pytest.fixture(scope="session")
def golden_modules():
    return load_golden("modules")


@pytest.fixture(scope="session")
def golden_phases():
    return load_golden("phases")


@pytest.fixture(scope="session")
def golden_greedy():
    return load_golden("greedy")
