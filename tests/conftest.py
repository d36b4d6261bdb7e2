import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference algorithm (test infrastructure only)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def mpf():
    """The product package (directory name has a '-', hence importlib)."""
    return importlib.import_module("mixed-precision_lu_factorization_amd")


@pytest.fixture(scope="session")
def ctx(mpf):
    """A live MPF context on cuda:0 -- GPU tests only.  Fails loudly if the HIP library is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    c = mpf.MPFContext(0)
    yield c
    c.close()


def bits16(t):
    """torch int16 tensor (fp16 bit patterns) -> numpy uint16."""
    return t.cpu().numpy().view(np.uint16)


GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
