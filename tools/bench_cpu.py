"""bench.py's cpu_baseline leg: the reference C path timed on this node's host cores (rank 0, N = 1 only).

The ONLY place the bench touches oracle/ (oracle/_ref = the real src_tb/dct32.c compiled in place when its prebuilt .so is present,
else the oracle's restatement): as the reported baseline and as the checker of the GPU batch, never as the thing measured."""
import ctypes
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.c_void_p


def host_cpu_facts():
    model, flags = "unknown", []
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            if line.startswith("flags") and not flags:
                have = set(line.split(":", 1)[1].split())
                flags = [f for f in ("avx2", "avx512f", "avx512bw", "avx512_vnni", "amx_int8") if f in have]
    except OSError:
        pass
    return model, flags


def usable_cpus():
    """(hardware threads this process may run on, CPUs the container's cgroup quota pays for).  A box can show
    256 hardware threads and grant 16 CPUs of quota: threads beyond the quota only add throttling."""
    hw = len(os.sched_getaffinity(0))
    quota = float(hw)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    return hw, quota


def best_thread_count(n_units, work, make_local):
    """Thread count for the all-cores figure = the CPUs the cgroup quota pays for (all hardware threads when there is
    no quota).  More threads than that only look faster in a short run (the quota is enforced per 100 ms period, so a
    burst borrows from the next period) and are throttled in a sustained one; the short trials at 2x / 4x / all
    threads are reported next to the figure for exactly that reason, not used."""
    hw, quota = usable_cpus()
    q = max(1, min(hw, int(quota + 0.999)))
    cands = sorted({c for c in (q, 2 * q, 4 * q, hw) if 1 <= c <= hw})
    trial = {}
    for c in cands:
        dt, _, _ = run_pinned(max(n_units // 4, c), c, work, make_local)
        trial[c] = max(n_units // 4, c) / dt
    return q, hw, quota, trial


def run_pinned(n_units, cores, work, make_local):
    """One pinned thread per core, one contiguous shard each.  Every thread first builds its OWN copy of
    its input shard and pre-touches its output shard (first touch => NUMA-local pages, no page faults in
    the timed region), then all start together.  Returns (seconds from the common start to the last
    finisher, list of per-thread outputs)."""
    bounds = np.linspace(0, n_units, cores + 1).astype(np.int64)
    cpus = sorted(os.sched_getaffinity(0))
    ready, go = threading.Barrier(cores + 1), threading.Barrier(cores + 1)
    ends = [0.0] * cores
    outs = [None] * cores

    def body(i):
        try:
            os.sched_setaffinity(0, {cpus[i % len(cpus)]})                # this thread only
        except OSError:
            pass
        b, e = int(bounds[i]), int(bounds[i + 1])
        loc_in, loc_out = make_local(b, e)
        outs[i] = loc_out
        ready.wait()
        go.wait()
        if e > b:
            work(loc_in, loc_out, e - b)
        ends[i] = time.perf_counter()

    ths = [threading.Thread(target=body, args=(i,)) for i in range(cores)]
    for th in ths:
        th.start()
    ready.wait()
    t0 = time.perf_counter()
    go.wait()
    for th in ths:
        th.join()
    return max(ends) - t0, outs, bounds


def native_port_rate(n, cores, make_local, outs_ref):
    """secondary figure: the restatement built -O3 -march=native ON THIS HOST; None when it cannot be built"""
    try:
        import glob
        import subprocess
        import tempfile
        so = os.path.join(tempfile.gettempdir(), "liborc_native_%d.so" % os.getpid())
        srcs = sorted(glob.glob(os.path.join(ROOT, "oracle", "*_oracle.c")))
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=gnu11", "-fPIC", "-shared", "-o", so] + srcs + ["-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        nat = ctypes.CDLL(so)

        def work_n(loc_in, loc_out, cnt):
            nat.orc_dct32_fwd_mt(P(loc_in.ctypes.data), P(loc_out.ctypes.data), ctypes.c_size_t(cnt), 1)
        dt_n, outs_n, _ = run_pinned(n, cores, work_n, make_local)
        os.unlink(so)
        return n / dt_n if all(np.array_equal(a, b) for a, b in zip(outs_n, outs_ref)) else None
    except Exception:
        return None


def cpu_baseline_dct(x_host, gpu_out_host):
    """Reference C path timed on the host cores (rank 0, N = 1).  Returns the
    cpu_baseline object; also checks the GPU output against it bit-for-bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _util import Oracle, Reference, ref_path

    orc = Oracle()
    n = x_host.shape[0]
    have_ref = os.path.exists(ref_path())
    ref = Reference() if have_ref else None

    def work(loc_in, loc_out, cnt):
        if have_ref:
            ref.lib.ref_dct32_fwd(P(loc_in.ctypes.data), P(loc_out.ctypes.data), ctypes.c_ulong(cnt))
        else:
            orc.lib.orc_dct32_fwd_mt(P(loc_in.ctypes.data), P(loc_out.ctypes.data), ctypes.c_size_t(cnt), 1)

    def make_local(b, e):
        loc_in = x_host[b:e].copy()
        return loc_in, np.zeros_like(loc_in)

    n1 = min(n, 32768)                                                    # single pinned thread, bounded sample, same code path
    dt1, _, _ = run_pinned(n1, 1, work, make_local)
    single = n1 / dt1
    cores, hw, quota, trial = best_thread_count(n, work, make_local)
    dt, outs, bounds = run_pinned(n, cores, work, make_local)
    exact = all(np.array_equal(outs[i], gpu_out_host[int(bounds[i]):int(bounds[i + 1])]) for i in range(cores))
    model, flags = host_cpu_facts()
    return {
        "value": n / dt, "unit": "blocks/s", "cores": cores, "kind": "reference" if have_ref else "port",
        "sample": "all %d blocks of the GPU batch (same inputs): %d pinned threads, one contiguous shard each, "
                  "thread-local input copy and pre-touched output (no page faults, NUMA-local), -O2" % (n, cores),
        "single_thread_blocks_per_s": single,
        "parallel_efficiency": (n / dt) / (min(cores, quota) * single),
        "host_hw_threads": hw, "container_cpu_quota": quota,
        "short_trials_blocks_per_s_by_threads": {str(k): v for k, v in trial.items()},
        "port_O3_march_native_all_cores_blocks_per_s": native_port_rate(n, cores, make_local, outs),
        "host_cpu": model, "host_cpu_flags": flags,
        "gpu_output_bit_exact_vs_cpu": exact,
    }, exact


def cpu_baseline_satd(dh, gpu_s):
    """the SATD port on the host cores over the first blocks of the GPU batch, and the GPU's costs against it"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _util import Oracle
    orc = Oracle()
    ns = dh.shape[0]

    def satd_work(loc_in, loc_out, cnt):
        orc.lib.orc_satd8x8_batch_mt(P(loc_in.ctypes.data), P(loc_out.ctypes.data), ctypes.c_size_t(cnt), 1)

    def mk(b, e):
        return dh[b:e].copy(), np.zeros(e - b, np.uint32)
    cores_s, _, _, _ = best_thread_count(ns, satd_work, mk)
    dt, outs, bounds = run_pinned(ns, cores_s, satd_work, mk)
    return {"value": ns / dt, "unit": "blocks/s", "cores": cores_s, "kind": "port",
            "sample": "first %d blocks of the GPU batch, %d pinned threads, pre-touched thread-local buffers" % (ns, cores_s),
            "gpu_output_bit_exact_vs_cpu": all(np.array_equal(outs[i], gpu_s[int(bounds[i]):int(bounds[i + 1])]) for i in range(len(outs)))}

