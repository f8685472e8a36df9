import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running")


@pytest.fixture(scope="session")
def oracle():
    from oracle import lk_oracle

    lk_oracle.lib()
    return lk_oracle


@pytest.fixture(scope="session")
def ml_small(oracle):
    return oracle.load_ml_small()


@pytest.fixture()
def rng():
    return np.random.default_rng(42)


@pytest.fixture(scope="session")
def gpu():
    "The HIP device; GPU tests FAIL (not skip) when the extension or device is missing."
    import torch

    from lkpy_amd import _native

    _native.require_gpu()
    assert torch.cuda.is_available()
    return torch.device("cuda:0")
