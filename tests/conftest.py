import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running test")


@pytest.fixture(scope="session")
def device():
    import torch

    return torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
