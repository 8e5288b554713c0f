import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU checker (oracle package), built on demand.  Test infrastructure only."""
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def fixtures():
    """The reference's own golden vectors (tests/golden/make_golden.py)."""
    return np.load(os.path.join(GOLDEN, "ref_fixtures.npz"))


@pytest.fixture(scope="session")
def generated():
    """Outputs of the unmodified reference on seeded inputs (tests/golden/make_golden.py)."""
    return np.load(os.path.join(GOLDEN, "ref_generated.npz"))


@pytest.fixture(scope="session")
def rd():
    """The product: richdem_amd over librdgpu.so.  No fallback -- import errors are failures."""
    import richdem_amd

    richdem_amd.lib()
    return richdem_amd


def gen_cases(generated):
    return sorted({k.split("/")[0] for k in generated.files})
