import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_NAME = "clifford-group-equivariant-simplicial-message-passing-networks_amd"
PKG_DIR = os.path.join(ROOT, PKG_NAME)
GOLDEN = os.path.join(ROOT, "tests", "golden")

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (hyphenated directory name, imported through importlib)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
