"""CPU: the C-ABI library builds for gfx950 without a GPU, loads, exports every
symbol include/x266hip.h declares, and its host-only entry points behave.  No
kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

import x266_amd
from _util import GOLDEN_DIR, ROOT


@pytest.fixture(scope="module")
def lib():
    x266_amd.build_library()
    return x266_amd.load_library()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "x266hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text)
    skip = {"defined", "if"}
    return sorted({n for n in names if n not in skip and not n.startswith("X266")})       # macros are not symbols


def test_header_symbols_exported(lib):
    names = _declared_symbols()
    for required in ("dct32_genNew", "dct32_getDiff", "dct32_getDct", "satd8x8_genNew", "satd8x8_getDiff",
                     "satd8x8_getSatd", "xDct32FwdBatchDev", "xDct32InvBatchDev", "xSatd8x8BatchDev",
                     "xDct32FwdBatch", "xSatd8x8Batch", "xHipCodecInit", "xHipCodecFree"):
        assert required in names
    for n in names:
        assert hasattr(lib, n), "include/x266hip.h declares %s but libx266hip.so does not export it" % n
    assert ctypes.c_int16.in_dll(lib, "g_t32") is not None


def test_exported_table_is_the_reference_matrix(lib, oracle):
    tab = np.ctypeslib.as_array((ctypes.c_int16 * 1024).in_dll(lib, "g_t32")).reshape(32, 32)
    assert np.array_equal(tab, oracle.table())


def test_pack_helpers_match_reference_sequence(lib, oracle):
    g = np.load(os.path.join(GOLDEN_DIR, "bdpi_dct32.npz"))
    for b in range(g["blocks"].shape[0]):
        for i in range(16):
            assert np.array_equal(x266_amd.pack_diff_rows(g["blocks"][b], 2 * i), g["diff_words"][b, i])
        got = [x266_amd.pack_dct_word(g["dcts"][b], 4 * i) for i in range(256)]
        assert got == [int(w) for w in g["dct_words"][b]]
        assert got[7] == oracle.pack_dct_word(g["dcts"][b], 28)


def test_no_cpu_fallback(lib):
    """Without a GPU the context cannot be created and nothing computes."""
    if lib.xHipDeviceCount() > 0:
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(x266_amd.X266Error):
        x266_amd.Codec(0)
    ctx = ctypes.c_void_p()
    assert lib.xHipCodecInit(ctypes.byref(ctx), 0) < 0 and not ctx.value
    # NULL context is rejected, not dereferenced
    assert lib.xDct32FwdBatchDev(None, None, None, 4, None) < 0
    assert lib.xSatd8x8Batch(None, None, None, 4) < 0


def test_product_does_not_link_the_oracle():
    """The shipped library must not reference anything under oracle/."""
    import subprocess
    out = subprocess.check_output(["ldd", x266_amd.lib_path()]).decode()
    assert "liborc" not in out and "x266ref" not in out
    syms = subprocess.check_output(["nm", "-D", x266_amd.lib_path()]).decode()
    assert "orc_" not in syms and "partialButterfly" not in syms
    for root, _, files in os.walk(os.path.join(ROOT, "x266_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "liborc" not in text and "oracle/" not in text.replace("oracle/ ", ""), f


def test_only_the_header_surface_is_exported(lib):
    """the library is linked into third-party executables -- nothing but the C ABI of
    include/x266hip.h (plus g_t32) may be visible (version script libx266hip.map)."""
    import subprocess
    declared = set(_declared_symbols()) | {"g_t32"}
    out = subprocess.check_output(["nm", "-D", "--defined-only", x266_amd.lib_path()]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert exported <= declared, "exported but not declared in include/x266hip.h: %s" % sorted(exported - declared)
    assert not [s for s in exported if s.startswith("_Z")]


def test_node_layer_without_a_gpu(lib):
    """The multi-GPU layer: planning functions are host-only; a node cannot be made without devices
    (no CPU path), NULL handles are rejected, and RCCL is NOT a link-time dependency of the drop-in."""
    import subprocess
    from x266_amd.node import Node, me_stripe_plan, shard_range
    assert shard_range(10, 0, 3) == (0, 4) and shard_range(10, 2, 3) == (7, 10)
    assert me_stripe_plan(2160, 64, 0, 8) == ((0, 34), (-64, 34 * 8 + 64))
    assert "rccl" not in subprocess.check_output(["ldd", x266_amd.lib_path()]).decode()
    assert lib.xNodeStreamFlush(None) < 0 and lib.xNodeStreamWait(None, 0) < 0
    assert lib.xHipNodeSelfTest(None) < 0 and lib.xNodeBatchScatterGather(None, 0, None, None, 4, 0) < 0
    if lib.xHipDeviceCount() == 0:
        with pytest.raises(x266_amd.X266Error):
            Node.single_process([0])


def test_product_transform_matrices_match_the_oracle_and_the_closed_form(lib, oracle):
    """The tables compiled into the product (x266_tables.hpp) -- not only the oracle's copies -- are the sub-matrices of
    g_t32 and the DST-VII closed form round(64 sqrt(N) sqrt(4/(2N+1)) sin(pi (2k+1)(n+1)/(2N+1)))."""
    lib.xTransformMatrix.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    g = oracle.table()
    for ttype in (0, 1):
        for n in (4, 8, 16, 32):
            m = np.zeros((n, n), np.int16)
            rc = lib.xTransformMatrix(ttype, n, m.ctypes.data)
            if ttype == 1 and n == 32:
                assert rc < 0
                continue
            assert rc == 0
            assert np.array_equal(m, oracle.transform_matrix(ttype, n))
            if ttype == 0:
                assert np.array_equal(m, g[:: 32 // n, :n])
            else:
                k, c = np.arange(n)[:, None], np.arange(n)[None, :]
                want = np.round(64 * np.sqrt(n) * np.sqrt(4.0 / (2 * n + 1)) * np.sin(np.pi * (2 * k + 1) * (c + 1) / (2 * n + 1)))
                assert np.array_equal(m, want.astype(np.int16))
    assert lib.xTransformMatrix(2, 8, None) < 0
