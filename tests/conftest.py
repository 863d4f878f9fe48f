import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "orb_slam2v2-1_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG + ".synth")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture
def hooks(pkg):
    """The test looks at intermediate stages (FAST candidates, quad-tree lists, blurred blocks ...): extractors it creates live in the
    DEVELOPER build of the library (liborbx_hip_dev.so: the same sources + the read-only hooks of include/orbx_dev.h).  Everything
    end-to-end or timed uses the product library, which exports no hook."""
    old = pkg.default_developer
    pkg.default_developer = True
    yield
    pkg.default_developer = old
