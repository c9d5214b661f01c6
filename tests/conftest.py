import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def torch_sees_the_gpu_before_the_library_does():
    """One GPU test (test_batch_size_is_halved...) fills the card through torch. torch bundles its own libamdhip64.so.7 (ROCm 7.0),
    libfcgpu.so links the system's (ROCm 7.2), and with one soname the process gets whichever is loaded FIRST for both. With
    torch's first everything works (bench.py's order); with the system's first torch later finds "No HIP GPUs". Running that
    test's file alone hit exactly that, so the order is fixed here, once per session, instead of depending on which test
    file happens to run first. No GPU: nothing to do."""
    try:
        import torch
    except ImportError:
        return
    if torch.cuda.is_available():
        torch.cuda.init()
