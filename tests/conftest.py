import os
import sys

import pytest

# the library looks at its CUVS_AMD_* kernel-selection / ablation switches only behind this gate (cuvs_amd/csrc/core.hip);
# the parity tests use them to force every code path
os.environ["CUVS_AMD_DEBUG_SWITCHES"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def res():
    """One cuvs resources handle for the GPU session."""
    import cuvs_amd

    return cuvs_amd.common.Resources()
