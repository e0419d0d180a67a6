import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    # The CPU oracle's convolutions (2 .. 1024 channels) crawl when torch's intra-op pool oversubscribes a 256-thread GPU host (bench.py
    # caps its cpu_baseline leg the same way): at most 32 threads for everything the suite computes on the host.
    import torch

    torch.set_num_threads(max(1, min(torch.get_num_threads(), 32)))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "autograd: the test records an autograd graph (everything else runs under torch.no_grad())")


@pytest.fixture(autouse=True)
def _inference_unless_autograd(request):
    """The product modules switch to the differentiable training composition whenever autograd is recording
    (syncfusion_amd/training.py).  The parity tests are about the inference ENGINE, as the reference's generation path runs it
    (main/generation.py:11 `@torch.no_grad()`), so every test runs under no_grad unless it is marked `autograd`."""
    import torch

    if request.node.get_closest_marker("autograd"):
        yield
    else:
        with torch.no_grad():
            yield


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
