import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/): the checker, never the thing under test."""
    from oracle import oracle as O  # noqa: N812
    O.build()
    return O


@pytest.fixture(scope="session")
def nsof_lib():
    """libnsof.so, built on demand (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build_native()
    import nsof
    return nsof


@pytest.fixture(scope="session")
def ctx(nsof_lib):
    c = nsof_lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def torch_dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch.device("cuda", 0)


def golden_path(name):
    return os.path.join(GOLDEN, name)


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)
