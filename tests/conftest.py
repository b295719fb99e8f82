import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A clean process factory, started BEFORE anything initialises the GPU: the multi-rank GPU tests get their worker
    # processes from this fork server (a process that has touched the GPU must not fork / exec others on the GPU boxes).
    import multiprocessing as mp
    from multiprocessing import forkserver
    mp.get_context("forkserver")
    forkserver.ensure_running()


@pytest.fixture(scope="session")
def hip():
    """The kernel C ABI; fails loudly (no fallback) when the library or the device is missing."""
    from prost_amd import _hip
    _hip.require_device()
    return _hip
