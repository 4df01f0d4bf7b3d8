import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The native pieces are git-ignored build products: bring them up to date before collection
    # (hipcc cross-compiles gfx950 without a GPU; make is a no-op when they are fresh) — and before
    # anything in this process touches the GPU.
    import __graft_entry__

    __graft_entry__.ensure_built()


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
