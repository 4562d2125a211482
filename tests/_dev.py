"""Device memory for the GPU tests through the C ABI only (xHipMalloc / xHipMemcpy*), and a ctypes view of the HIP
runtime THE LIBRARY loaded -- no torch in a test process: the library is validated on the runtime it is built for
(ROCm's libamdhip64 / librccl), which is what a plain-C host gets.

`Dev(codec)` hands out small tensor-like objects with the handful of methods the tests use (data_ptr, numel, stride,
cpu().numpy(), clone) so that test bodies read as they did when torch supplied the memory."""
import ctypes
import re

import numpy as np


def loaded_libraries(pattern):
    """paths of the shared objects mapped into this process whose file name matches `pattern` (a regex)"""
    seen = []
    for line in open("/proc/self/maps"):
        parts = line.split()
        if len(parts) >= 6 and re.search(pattern, parts[5].rsplit("/", 1)[-1]) and parts[5] not in seen:
            seen.append(parts[5])
    return seen


def hip_runtime():
    """ctypes handle of the libamdhip64 this process runs on (the library's own: there must be exactly one)"""
    import x266_amd
    x266_amd.load_library()
    libs = loaded_libraries(r"^libamdhip64\.so")
    assert len(libs) == 1, "expected exactly one HIP runtime in the process, found %r" % (libs,)
    return ctypes.CDLL(libs[0])


class DevArray:
    def __init__(self, codec, shape, dtype):
        self.codec = codec
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self._n = int(np.prod(self.shape)) if self.shape else 1
        self.buf = codec.alloc(max(self._n * self.dtype.itemsize, 16))

    def data_ptr(self):
        return self.buf.ptr

    def numel(self):
        return self._n

    def stride(self, axis):
        s = 1
        for d in self.shape[axis + 1:]:
            s *= d
        return s

    def numpy(self):
        self.codec.stream_sync()
        return self.buf.download(self.dtype, self._n).reshape(self.shape)

    def cpu(self):
        return self

    def clone(self):
        c = DevArray(self.codec, self.shape, self.dtype)
        c.buf.upload(self.numpy())
        return c

    def upload(self, a):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.size == self._n
        self.buf.upload(a)


class Dev:
    """factory bound to one context (= one device)"""

    def __init__(self, codec):
        self.codec = codec

    def from_numpy(self, a):
        a = np.ascontiguousarray(a)
        t = DevArray(self.codec, a.shape, a.dtype)
        t.buf.upload(a)
        return t

    def zeros(self, shape, dtype):
        t = DevArray(self.codec, shape, dtype)
        t.buf.upload(np.zeros(t._n, t.dtype))
        return t

    def empty(self, shape, dtype):
        return DevArray(self.codec, shape, dtype)

    def empty_like(self, t):
        return DevArray(self.codec, t.shape, t.dtype)

    def random_u8(self, shape, seed):
        return self.from_numpy(np.random.RandomState(seed).randint(0, 256, shape).astype(np.uint8))

    def synchronize(self):
        self.codec.stream_sync()

    @staticmethod
    def equal(a, b):
        return a.shape == b.shape and np.array_equal(a.numpy(), b.numpy())
