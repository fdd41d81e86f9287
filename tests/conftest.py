"""pytest configuration: registers the `gpu` marker and puts the harness modules on sys.path."""
import os
import sys

import pytest

try:   # before libmvs_hip.so is loaded: the library must bind to the HIP runtime torch brings along (same soname);
    import torch  # noqa: F401  loaded the other way round, torch's own runtime finds no device ("No HIP GPUs are available")
except ImportError:
    pass

# the sweep's timing-experiment flag bits (flags >> 8: debug switches, forced plane-split counts) are masked off by the library unless
# this is set; several parity tests drive the kernels through them (bit-identical results are the point of those tests)
os.environ.setdefault("MVS_DEBUG_FLAGS", "1")
# the master switch of every environment hook of the library (csrc/hooks.hpp): without it the library reads no
# other variable.  Contexts and communicators take their snapshot when they are created.
os.environ.setdefault("MVS_TEST_HOOKS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """no test may hang the box it runs on (a hung GPU box is lost to everybody): with pytest-timeout installed every test gets a
    generous limit -- the slowest takes about a minute"""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            # GPU tests: the watchdog-thread method (it ends the process): a signal cannot interrupt a call stuck inside the HIP runtime
            item.add_marker(pytest.mark.timeout(900, method="thread") if item.get_closest_marker("gpu") else pytest.mark.timeout(900))


@pytest.fixture(scope="session")
def oracle():
    import orc
    return orc.load()
