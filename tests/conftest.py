import os
import sys

import pytest

# No torch in a test process: device memory comes from the C ABI (tests/_dev.py), so the library is exercised on the HIP
# runtime and RCCL it is built for and ships on (ROCm's /opt/rocm copies, what a plain-C host gets) -- the torch wheel bundles
# an older libamdhip64 / librccl with the same SONAMEs, and whichever copy a process loads first serves everything in it
# (tests/test_gpu_runtime.py asserts there is exactly one, and which).  Only tests/test_shard_gloo.py (CPU, gloo) and the
# N > 1 control plane of bench.py import torch, the latter AFTER the library.

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_collection_finish(session):
    # no session imports torch into pytest's own process (see above): the gloo tests use it in their worker processes only
    assert "torch" not in sys.modules, "collecting the tests imported torch: its bundled HIP runtime / RCCL would serve the test process"


def pytest_sessionfinish(session, exitstatus):
    assert "torch" not in sys.modules, "a test imported torch into pytest's own process (see tests/conftest.py)"


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
