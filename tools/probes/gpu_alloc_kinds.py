#!/usr/bin/env python3
"""Developer probe (round 5): does the KIND of device allocation change what the 1 : 1 stream reaches?  hipMalloc (what xHipMalloc uses) against hipExtMallocWithFlags
(default / fine-grained / uncached / contiguous), hipMallocAsync from the default pool, and one pass of the runtime's own device-to-device copy (hipMemcpyDtoDAsync)
for comparison: the copy stream, the forward DCT32 and the read probe on each."""
import ctypes, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
from tests._dev import loaded_libraries
cd = x266_amd.Codec(0)
hip = ctypes.CDLL(loaded_libraries("libamdhip64.so")[0])
P = ctypes.c_void_p
hip.hipMalloc.argtypes = [ctypes.POINTER(P), ctypes.c_size_t]
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(P), ctypes.c_size_t, ctypes.c_uint]
hip.hipMallocAsync.argtypes = [ctypes.POINTER(P), ctypes.c_size_t, P]
hip.hipMemcpyDtoDAsync.argtypes = [P, P, ctypes.c_size_t, P]
hip.hipFree.argtypes = [P]
n = 1 << 20
nbytes = n * 2048
N = 16
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=5):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
def alloc(kind):
    p = P()
    if kind == "hipMalloc": rc = hip.hipMalloc(ctypes.byref(p), nbytes)
    elif kind == "async pool": rc = hip.hipMallocAsync(ctypes.byref(p), nbytes, None)
    else: rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, {"ext default": 0x0, "ext fine-grained": 0x1, "ext uncached": 0x3, "ext contiguous": 0x4}[kind])
    return p.value if rc == 0 else None
for rnd in range(2):
    for kind in ("hipMalloc", "ext default", "ext fine-grained", "ext uncached", "ext contiguous", "async pool"):
        x, z = alloc(kind), alloc(kind)
        if not x or not z:
            print("%-17s: allocation refused" % kind); continue
        cd.fill_residual_dev(x, n * 1024, 0x266); cd.stream_sync()
        t = (timed(lambda: cd.mem_ceiling_dev(0, x, z, nbytes)), timed(lambda: cd.dct32_fwd_dev(x, z, n)), timed(lambda: cd.mem_ceiling_dev(3, x, z, nbytes)),
             timed(lambda: hip.hipMemcpyDtoDAsync(z, x, nbytes, None)))
        print("%-17s: copy stream %.4f ms (%.2f TB/s)  forward DCT32 %.4f  read probe %.4f (%.2f TB/s)  hipMemcpyDtoD %.4f (%.2f TB/s)" % (kind, t[0], 2 * nbytes / t[0] / 1e9, t[1], t[2], nbytes / t[2] / 1e9, t[3], 2 * nbytes / t[3] / 1e9), flush=True)
