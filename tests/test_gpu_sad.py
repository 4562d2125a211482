"""GPU (-m gpu): batched SAD -- the reference's own known answer
(64x64 -> 344807, riscv/programs/benchmarks/sad) and random batches against numpy."""
import os

import numpy as np
import pytest

from _util import GOLDEN_DIR, splitmix64

pytestmark = pytest.mark.gpu


def test_reference_known_answer(codec):
    g = np.load(os.path.join(GOLDEN_DIR, "sad64.npz"))
    assert int(g["sad"][0]) == 344807
    assert codec.sad(64, g["a"], g["b"]).tolist() == [344807]
    # the same data cut into smaller blocks must add up to the same total
    for edge in (4, 8, 16, 32):
        a = g["a"].reshape(64 // edge, edge, 64 // edge, edge).transpose(0, 2, 1, 3).reshape(-1, edge * edge)
        b = g["b"].reshape(64 // edge, edge, 64 // edge, edge).transpose(0, 2, 1, 3).reshape(-1, edge * edge)
        assert int(codec.sad(edge, a, b).sum()) == 344807


def test_per_call_twin_of_the_benchmark_function(codec):
    """x266_sad(a, b, n): same arguments and result as sad() of riscv/programs/benchmarks/sad/sad.c:28-39."""
    import ctypes
    L = codec.L
    L.x266_sad.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.x266_sad.restype = ctypes.c_int
    g = np.load(os.path.join(GOLDEN_DIR, "sad64.npz"))
    a, b = np.ascontiguousarray(g["a"], np.uint8).ravel(), np.ascontiguousarray(g["b"], np.uint8).ravel()
    assert L.x266_sad(a.ctypes.data, b.ctypes.data, 64) == 344807                     # the benchmark's own check value
    for n in (4, 8, 16, 32):
        x, y = a[: n * n], b[: n * n]
        assert L.x266_sad(x.ctypes.data, y.ctypes.data, n) == int(np.abs(x.astype(np.int64) - y.astype(np.int64)).sum())
    assert L.x266_sad(a.ctypes.data, b.ctypes.data, 5) == -1 and L.x266_sad(None, b.ctypes.data, 8) == -1


@pytest.mark.parametrize("edge", [4, 8, 16, 32, 64])
@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 63, 64, 65, 1000, 4097])
def test_random_batches(codec, edge, n):
    r = splitmix64(edge * 1000 + n, 0, 2 * n * edge * edge)
    a = (r[: n * edge * edge] & np.uint64(0xFF)).astype(np.uint8).reshape(n, -1)
    b = ((r[n * edge * edge:] >> np.uint64(17)) & np.uint64(0xFF)).astype(np.uint8).reshape(n, -1)
    want = np.abs(a.astype(np.int32) - b.astype(np.int32)).sum(axis=1).astype(np.uint32)
    assert np.array_equal(codec.sad(edge, a, b), want)


def test_extremes_and_errors(codec):
    a = np.zeros((5, 4096), np.uint8)
    b = np.full((5, 4096), 255, np.uint8)
    assert codec.sad(64, a, b).tolist() == [255 * 4096] * 5
    buf = codec.alloc(8192)
    assert codec.L.xSadBatchDev(codec.ctx, 12, buf.ptr, buf.ptr, buf.ptr, 1, None) < 0
    assert codec.L.xSadBatchDev(codec.ctx, 8, None, buf.ptr, buf.ptr, 1, None) < 0
    assert codec.L.xSadBatchDev(codec.ctx, 8, None, None, None, 0, None) == 0


@pytest.mark.parametrize("edge", [8, 64])
def test_sad_batches_beyond_4_gib(codec, edge):
    """Maximum sizes: 4 GiB + a ragged tail per input; samples at the start, across the 2^32-byte boundary and at the end against numpy."""
    n = ((1 << 32) // (edge * edge)) + 5
    da, db, dout = codec.alloc(n * edge * edge), codec.alloc(n * edge * edge), codec.alloc(n * 4)
    codec.fill_residual_dev(da.ptr, n * edge * edge // 2, 0x5AD)                 # any bytes
    codec.fill_residual_dev(db.ptr, n * edge * edge // 2, 0x5AE)
    codec.sad_dev(edge, da.ptr, db.ptr, dout.ptr, n)
    codec.stream_sync()

    def fetch(buf, byte_off, count, dtype):
        out = np.empty(count, dtype)
        codec._check(codec.L.xHipMemcpyD2H(codec.ctx, out.ctypes.data, buf.ptr + byte_off, out.nbytes), "D2H")
        return out

    for first, count in [(0, 9), (n - 5 - 4, 9), (n - 3, 3)]:                    # block n - 5 starts at byte 2^32
        a = fetch(da, first * edge * edge, count * edge * edge, np.uint8).reshape(count, -1).astype(np.int64)
        b = fetch(db, first * edge * edge, count * edge * edge, np.uint8).reshape(count, -1).astype(np.int64)
        assert np.array_equal(fetch(dout, first * 4, count, np.uint32), np.abs(a - b).sum(axis=1).astype(np.uint32)), (edge, first)
