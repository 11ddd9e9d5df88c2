import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)


@pytest.fixture(autouse=True)
def _main_stream_param_grads():
    """model._setup_training switches filter/bias gradients to their own stream (joined inside train_step); tests
    that call loss.backward() themselves and read gradients right away get the plain single-stream order."""
    yield
    ops = sys.modules.get("vnet_tensorflow_amd.ops")
    if ops is not None:
        ops.join_param_grad_stream()
        ops.set_param_grad_stream(False)


@pytest.fixture
def lib_option():
    """lib_option(name, value): vnet_set_option for the duration of a test (the library reads its switches once; tests flip them here)."""
    from vnet_tensorflow_amd import _lib
    saved = []

    def setopt(name, value):
        saved.append((name, _lib.set_option(name, float(value))))
    yield setopt
    for name, prev in reversed(saved):
        _lib.set_option(name, prev)
