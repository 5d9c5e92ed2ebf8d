import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure; see oracle/glb_oracle.c)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def engine():
    import torch
    import genlm_backend_amd
    from genlm_backend_amd.engine import HipEngine
    assert torch.cuda.is_available(), "gpu-marked test without a GPU"
    # the contract the oracle restates bit for bit; 16-bit rows under the hardware exponential (the product's default for
    # them) are tests/test_step_hw_gpu.py's, per call
    return HipEngine("cuda:0", contract="poly")
