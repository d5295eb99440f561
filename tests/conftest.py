"""pytest configuration: marker registration and shared fixtures.

`-m "not gpu"` : oracle vs the reference's golden vectors, host logic, and the
                 C-ABI library's load/export surface (no GPU compute).
`-m gpu`       : parity tests proper; every call goes through the C ABI.
"""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs through the HIP C-ABI)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("bitnet-rs_amd")


@pytest.fixture(scope="session")
def hip(pkg):
    """Loaded + initialised C-ABI library wrapper; GPU tests only."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    lib = pkg.load()
    lib.init(0)
    return lib
