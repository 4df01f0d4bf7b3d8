import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The native pieces are git-ignored build products: make sure they exist before collection
    # (hipcc cross-compiles gfx950 without a GPU; a no-op when they are up to date).
    import subprocess

    if not os.path.exists(os.path.join(ROOT, "dream_gnn_amd", "libdgmi.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "dream_gnn_amd", "csrc"), "-j4"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libdgmi_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): builds oracle/libdgmi_oracle.so on first use."""
    from oracle import oracle as o

    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
