import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_sessionstart(session):
    """Some GPU tests move halos as torch CUDA tensors next to the C-ABI library.  torch ships its own libamdhip64; whichever
    HIP runtime is loaded FIRST in a process is the one that owns the GPU, and torch cannot initialise after /opt/rocm's copy
    has been loaded by libidocp_hip.so.  So on a GPU box torch's runtime is brought up before any test loads the library
    (the library then binds to the same runtime).  Without a GPU this does nothing."""
    try:
        import torch
        if torch.cuda.device_count() > 0 and torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
