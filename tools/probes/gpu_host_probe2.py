#!/usr/bin/env python3
"""Developer probe (round 4): the host-pointer API (xDct32FwdBatch: three staging slots) against what the PCIe link gives --
one-direction and both-direction plain copies from pinned memory -- with the link's negotiated generation / width from sysfs."""
import ctypes, glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import x266_amd
for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        if open(dev + "/vendor").read().strip() != "0x1002":
            continue
        print(dev, {k: open(dev + "/" + k).read().strip() for k in ("current_link_speed", "current_link_width", "max_link_speed", "max_link_width") if os.path.exists(dev + "/" + k)})
    except OSError:
        pass
cd = x266_amd.Codec(0)
print("X266HIP_HOST_REGISTER =", os.environ.get("X266HIP_HOST_REGISTER"))
n = 1 << 17
nbytes = n * 2048
hp_in, hp_out = torch.empty(nbytes, dtype=torch.uint8).pin_memory(), torch.empty(nbytes, dtype=torch.uint8).pin_memory()
d_a, d_b = torch.empty(nbytes, dtype=torch.uint8, device="cuda"), torch.empty(nbytes, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def t(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


dt = t(lambda: d_a.copy_(hp_in, non_blocking=True)); print("link H2D alone (pinned, one hipMemcpyAsync of 256 MiB): %.1f GB/s" % (nbytes / dt / 1e9))
dt = t(lambda: hp_out.copy_(d_b, non_blocking=True)); print("link D2H alone: %.1f GB/s" % (nbytes / dt / 1e9))


def both():
    with torch.cuda.stream(s1):
        d_a.copy_(hp_in, non_blocking=True)
    with torch.cuda.stream(s2):
        hp_out.copy_(d_b, non_blocking=True)
dt = t(both); print("link H2D + D2H at once: %.1f GB/s each way" % (nbytes / dt / 1e9))
x = np.random.default_rng(1).integers(-255, 256, size=(n, 1024), dtype=np.int16)
z = np.ones_like(x)
cd.dct32_fwd(x[:1024])
for rep in range(4):
    t0 = time.perf_counter(); rc = cd.L.xDct32FwdBatch(cd.ctx, ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(z.ctypes.data), n); dt = time.perf_counter() - t0
    print("xDct32FwdBatch pageable (touched) n=%d: %.1f ms  %.3e blocks/s  %.1f GB/s each way" % (n, dt * 1e3, n / dt, nbytes / dt / 1e9), flush=True)
ref = z.copy()
xp = torch.from_numpy(x).pin_memory(); zp = torch.empty_like(xp).pin_memory()
for rep in range(4):
    t0 = time.perf_counter(); rc = cd.L.xDct32FwdBatch(cd.ctx, ctypes.c_void_p(xp.data_ptr()), ctypes.c_void_p(zp.data_ptr()), n); dt = time.perf_counter() - t0
    print("xDct32FwdBatch pinned n=%d: %.1f ms  %.3e blocks/s  %.1f GB/s each way" % (n, dt * 1e3, n / dt, nbytes / dt / 1e9), flush=True)
assert rc == 0 and np.array_equal(zp.numpy(), ref)
