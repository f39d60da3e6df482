import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("radiativetransfer-sos_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ctypes
    oracle_ctypes.lib()
    return oracle_ctypes


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the HIP path has no CPU fallback)")
    pkg.capi.lib()  # raises loudly if libsosgpu.so is missing
    return pkg
