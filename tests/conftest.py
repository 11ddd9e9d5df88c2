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
