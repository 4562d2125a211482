import ctypes, os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
L = ctypes.CDLL(os.path.join(R, "x266_amd", "libx266hip.so"))
L.dct32_getDct.restype = ctypes.c_ulonglong
for name, fn in (("dct32_genNew", L.dct32_genNew), ("satd8x8_genNew", L.satd8x8_genNew)):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    N = 2000
    for _ in range(N): fn()
    dt = (time.perf_counter() - t0) / N * 1e6
    print("%s: %.1f us per call" % (name, dt))
