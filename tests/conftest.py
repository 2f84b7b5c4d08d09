import os
import sys

import pytest

# PyTorch first: it ships its own libamdhip64, and the process should hold ONE HIP runtime.  libamrdemod.so binds to
# whatever libamdhip64 is loaded already; loaded the other way round (the system ROCm runtime first, torch second) the
# torch-stream test cannot share streams and device pointers with the decoder.
try:
    import torch  # noqa: F401
except ImportError:   # the product does not need torch; only the torch-plumbing and multi-process tests do
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def amr_lib():
    """Loads libamrdemod.so (must have been built by __graft_entry__.build())."""
    from rtlamr_amd import _lib
    return _lib.lib()
