import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """Initialise the HIP library on device 0; fails loudly when there is no usable GPU."""
    # PyTorch ships its own copy of the HIP runtime: when both live in one process torch has to
    # be loaded first, or its later initialisation finds no device
    import torch
    torch.cuda.is_available()
    from raytracer_amd import device
    device.init(0)
    return device
