import os
import sys

import pytest

# The library looks at its CUVS_AMD_* kernel-selection / ablation switches only behind the gate CUVS_AMD_DEBUG_SWITCHES=1, and
# only when a handle is created (cuvs_amd/csrc/core.hip). The gate is NOT set for the session: the session handle `res` and
# every handle a test creates without touching a switch run the PRODUCTION configuration. A test that sets a CUVS_AMD_* switch
# through `monkeypatch.setenv` gets the gate with it (the override below) - for the comparator handle it is about to create.
os.environ.pop("CUVS_AMD_DEBUG_SWITCHES", None)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_NOT_SWITCHES = ("CUVS_AMD_DEBUG_SWITCHES", "CUVS_AMD_SHM_TIMEOUT_S", "CUVS_AMD_SHM_DIR", "CUVS_AMD_TABLE_LOG")


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with the gate of the library's debug switches set together with the first switch a test sets."""
    plain_setenv = monkeypatch.setenv

    def setenv(name, value, prepend=None):
        if name.startswith("CUVS_AMD_") and name not in _NOT_SWITCHES:
            plain_setenv("CUVS_AMD_DEBUG_SWITCHES", "1")
        plain_setenv(name, value, prepend)

    monkeypatch.setenv = setenv
    return monkeypatch


@pytest.fixture(scope="session")
def res():
    """One cuvs resources handle for the GPU session."""
    import cuvs_amd

    return cuvs_amd.common.Resources()
