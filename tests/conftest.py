import os
import sys

import pytest

# Some GPU tests use torch for device memory next to the library.  The torch wheel bundles its own HIP runtime with the
# same SONAME as ROCm's, and torch only finds the GPU behind its own copy: whichever test
# runs first, torch's copy has to be the one the process loads.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    from _util import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The real reference golden models (oracle/_ref); absent outside the build
    container unless the prebuilt .so travelled with the snapshot."""
    from _util import Reference, ref_path
    if not os.path.exists(ref_path()):
        pytest.skip("oracle/_ref/libx266ref.so not built (needs /root/reference)")
    return Reference()


@pytest.fixture(scope="session")
def codec():
    """GPU context through the C ABI.  No fallback: a missing library or device
    is a hard failure for -m gpu tests."""
    import x266_amd
    return x266_amd.Codec(0)
