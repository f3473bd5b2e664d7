import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_checker():
    """The CPU oracle exposed with the op signatures of dskd_amd.native (tests only)."""
    from oracle.checker import OracleChecker
    return OracleChecker()


@pytest.fixture()
def cpu_ops(oracle_checker):
    """Let host-logic tests run the model on CPU tensors by injecting the oracle as the
    checker implementation of the hot-path ops.  Removed again after the test."""
    from dskd_amd import native
    native.install_cpu_checker(oracle_checker)
    yield oracle_checker
    native.install_cpu_checker(None)
