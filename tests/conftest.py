import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import bind
    bind.build(ref=None)
    return bind


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """On a GPU box PyTorch brings its own HIP / HSA runtime (torch/lib/libamdhip64.so) next to the system one libsonde_hip.so links against.
    Both can live in one process, but only when PyTorch's is initialised first: a test process that ran a scanner through libsonde_hip before
    the first torch.cuda call then got `No HIP GPUs are available` from PyTorch (seen with tests/test_gpu_scan.py ahead of tests/test_gpu_chan.py).
    bench.py has that order anyway (Dist() selects the device before any engine exists); the tests get it here, whatever their order."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
