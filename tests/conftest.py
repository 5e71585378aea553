import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # torch bundles its own HIP runtime; when it is initialised AFTER the library's
    # (system ROCm) it reports "No HIP GPUs are available".  GPU tests that hand
    # library memory to torch need torch first, so initialise it up-front on a GPU box.
    try:
        import torch

        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:
        pass
