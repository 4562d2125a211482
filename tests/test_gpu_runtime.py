"""GPU: the test process runs on the runtime the library SHIPS on.

libx266hip.so is built by ROCm 7.2's hipcc against /opt/rocm's libamdhip64, and its node layer dlopens librccl.so.1.  The torch
wheel bundles older copies of both under the same SONAMEs, and the copy a process loads first serves everything in it -- rounds
1-4 imported torch in conftest.py, so 509 tests validated the library on HIP 7.0 / RCCL 2.26.  No test process imports torch
any more (device memory: tests/_dev.py); this file asserts what the process really has."""
import ctypes
import sys

import numpy as np
import pytest

from _dev import hip_runtime, loaded_libraries

pytestmark = pytest.mark.gpu


def test_exactly_one_hip_runtime_and_it_is_rocms(codec):
    assert "torch" not in sys.modules, "a GPU test imported torch: its bundled HIP runtime would serve the whole process"
    libs = loaded_libraries(r"^libamdhip64\.so")
    assert len(libs) == 1 and "/torch/" not in libs[0], libs
    v = ctypes.c_int()
    assert hip_runtime().hipRuntimeGetVersion(ctypes.byref(v)) == 0
    assert v.value >= 70200000, "HIP runtime %d: the library is built for ROCm 7.2" % v.value
    print("\nHIP runtime %d from %s" % (v.value, libs[0]))


def test_the_node_layer_talks_to_rocms_rccl():
    from x266_amd.node import Node
    node = Node.for_rank(0, 0, 1, Node.unique_id())
    node.self_test()
    ver, path = Node.rccl_info()
    node.close()
    libs = loaded_libraries(r"^librccl")
    assert len(libs) == 1 and "/torch/" not in libs[0] and "rccl_model" not in libs[0], libs
    assert ver >= 22700, (ver, path)                       # ROCm 7.2 ships RCCL 2.27.x; the torch wheel's copy is 2.26
    print("\nRCCL %d from %s" % (ver, libs[0]))


def test_pinned_block_outlives_the_array_it_was_handed_out_as(codec):
    """views, slices and reshapes of a host_alloc array keep the pinned block alive; the block goes with the last
    of them, and a context closed meanwhile goes with the last block."""
    import gc
    import x266_amd
    cd = x266_amd.Codec(0)
    a = cd.host_alloc((64, 1024), np.int16)
    a[:] = np.arange(64 * 1024, dtype=np.int16).reshape(64, 1024)
    view, flat = a[3:5], a.reshape(-1)[1024:2048]
    assert cd._pinned_live == 1
    del a
    gc.collect()
    assert cd._pinned_live == 1                            # two views alive: nothing freed
    out = cd.dct32_fwd(view)                               # the host-pointer call reads through a VIEW of the pinned block
    assert out.shape == (2, 1024) and np.array_equal(flat, np.arange(1024, 2048, dtype=np.int16))
    with pytest.warns(ResourceWarning, match="1 pinned host block"):
        cd.close()                                         # deferred: a block is alive -- said aloud ...
    assert cd.close_deferred and cd.ctx is None
    with pytest.raises(x266_amd.X266Error, match="closed"):
        cd.dct32_fwd(flat)                                 # ... and the codec takes no new calls meanwhile
    assert np.array_equal(flat, np.arange(1024, 2048, dtype=np.int16))   # the pinned memory itself is still there
    del view
    gc.collect()
    assert cd._pinned_live == 1 and cd.close_deferred
    del flat
    gc.collect()
    assert cd._pinned_live == 0 and cd.ctx is None and not cd.close_deferred   # last view gone: block freed, then the context
    cd.close()                                             # idempotent
