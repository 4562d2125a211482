#!/usr/bin/env python3
"""bench.py -- throughput of the x266 DCT32 / SATD hot path on MI355X.

Contract (one JSON line on stdout from rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  (plain `python bench.py --gpus N` without WORLD_SIZE in the environment spawns exactly that itself)

Workload = BASELINE.json configs[1]: 1,048,576 synthetic 9-bit residual blocks of 32x32 int16 per GPU, resident in HBM
before the timed region (values a-b, a,b uniform bytes -- the reference's stimulus distribution, src_tb/dct32.c:191-193 --
from SplitMix64 seed 0x266).  A "step" is one forward 2-D DCT32 pass over the batch (xDct32FwdBatchDev through the C ABI).
`value` is whole-job forward blocks/s over all ranks; every other leg (inverse, fused forward+inverse, the 8x8 SATD residual
batch of 2^24 blocks, motion search, the transform set, the 8K frame stream of configs[4], ...) is one function below,
measured the same way, and lands under "also".

The process runs on the runtime the library ships on: device memory, streams and events come from the C ABI (xHipMalloc,
xHipMemcpy*, xHipEvent*) and, for the two raw link copies of the host-API leg, from the HIP runtime libx266hip.so itself
loaded (ctypes) -- no torch in the single-GPU path.  With N > 1, torch.distributed is the CONTROL plane only, on gloo,
imported after the library: barrier, max-over-ranks time, checksum sum, the broadcast of the node's RCCL id; the data path
(RCCL send/recv groups of the node layer, x266_amd/csrc/x266hip_node.cpp) then talks to ROCm's librccl, not the older copy
the torch wheel bundles.  `hip_runtime` / `rccl_by_rank` in the line say which libraries the process really had.

How every leg is timed (`Bench.timed_leg`): its own clock pre-warm (the chip needs ~50 ms of load to reach steady clocks,
profiles/r01_clock_warmup.txt), W untimed launches, then K launches between barrier + device synchronize on both sides; a HIP
event is recorded ON THE LAUNCHING STREAM before every one of those K launches and after the last, so the kernel durations
come from the very launches whose wall-clock is `ms_per_step`.  Fractions use `kernel_ms` = the 10 % trimmed mean of those K
durations (a profiler's or the host's hiccup on 1 launch in 20 must not move a fraction by 2x: round 4's sad_8x8 row);
the plain mean and the median are reported next to it.

"roofline": algorithmic bytes per launch (4096 B per DCT block, 132 B per SATD block) / kernel_ms,
against the 8 TB/s HBM3E peak.  "cpu_baseline": the reference C path on this node's host cores (oracle/_ref = the real
src_tb/dct32.c when its prebuilt .so is present, else the oracle's restatement), rank 0 at N = 1 only.  The oracle is used
there and nowhere in the product.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_BYTES_PER_S = 8.0e12          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
XGMI_LINK_BYTES_PER_S = 153e9          # per direction, per peer link
DCT_BLOCKS_PER_GPU = 1 << 20           # BASELINE configs[1]
SATD_BLOCKS_PER_GPU = 1 << 24          # 2 GiB of 8x8 residual blocks
DCT_BYTES_PER_BLOCK = 4096             # 2048 read + 2048 written   
SATD_BYTES_PER_BLOCK = 132             # 128 read + 4 written
DCT_SEED, SATD_SEED = 0x266, 0x267
PREWARM_SECONDS = 0.08
P = ctypes.c_void_p


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dct-blocks", type=int, default=DCT_BLOCKS_PER_GPU)
    ap.add_argument("--satd-blocks", type=int, default=SATD_BLOCKS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="headline leg only")
    ap.add_argument("--stream8k", type=int, default=200, metavar="FRAMES",
                    help="frames of the BASELINE configs[4] leg (7680x4320 stream through the node layer); 0 skips it")
    ap.add_argument("--node-timeout", type=float, default=240.0,
                    help="seconds the node-layer legs (RCCL) may take before the JSON line is printed without them")
    ap.add_argument("--no-me", action="store_true", help="skip the motion-search legs")
    ap.add_argument("--no-host-api", action="store_true", help="skip the PCIe-inclusive host-pointer leg")
    ap.add_argument("--no-transform-set", action="store_true", help="skip the transform-set / front-end / intra legs")
    ap.add_argument("--no-autotune", action="store_true", help="skip the default-vs-autotuned leg (profiled runs: its candidate launches would mix shapes into the per-kernel averages)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc passes that measure roofline.traffic (replay profiles/traffic.json instead)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)   # the profiled child of the live traffic passes
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# the HIP runtime of this process (the copy libx266hip.so loaded), through ctypes
# ------------------------------------------------------------------------------------------------
def loaded_libraries(prefix):
    """paths of the shared objects mapped into this process whose file name starts with `prefix`"""
    seen = []
    for line in open("/proc/self/maps"):
        parts = line.split()
        if len(parts) >= 6 and parts[5].rsplit("/", 1)[-1].startswith(prefix) and parts[5] not in seen:
            seen.append(parts[5])
    return seen


class HipRuntime:
    """what bench.py needs of HIP beyond the C ABI: device synchronize, the PCI address, raw async copies for the link's own rate"""

    def __init__(self):
        self.paths = loaded_libraries("libamdhip64.so")
        if not self.paths:
            raise SystemExit("libx266hip.so did not bring a HIP runtime into the process")
        self.lib = L = ctypes.CDLL(self.paths[0])
        L.hipMemcpyAsync.argtypes = [P, P, ctypes.c_size_t, ctypes.c_int, P]
        L.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(P), ctypes.c_uint]
        L.hipStreamSynchronize.argtypes = [P]
        L.hipStreamDestroy.argtypes = [P]
        L.hipDeviceGetPCIBusId.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: hipError %d" % (what, rc))

    def device_sync(self):
        self.check(self.lib.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def version(self):
        v = ctypes.c_int()
        self.check(self.lib.hipRuntimeGetVersion(ctypes.byref(v)), "hipRuntimeGetVersion")
        return v.value

    def pci_bus_id(self, device):
        buf = ctypes.create_string_buffer(64)
        return buf.value.decode().lower() if self.lib.hipDeviceGetPCIBusId(buf, 64, device) == 0 else None

    def stream_create(self):
        s = P()
        self.check(self.lib.hipStreamCreateWithFlags(ctypes.byref(s), 1), "hipStreamCreateWithFlags")   # hipStreamNonBlocking
        return s

    def memcpy_async(self, dst, src, nbytes, kind, stream):
        self.check(self.lib.hipMemcpyAsync(P(dst), P(src), nbytes, kind, stream), "hipMemcpyAsync")


# ------------------------------------------------------------------------------------------------
# roofline.traffic, live: HBM bytes of one headline launch from the PMC counters, collected as
# MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes with --kernel-trace only
# (never with sys / hip / hsa traces), units KB.  gfx950's FETCH_SIZE counts the 128-byte requests of 16 B-per-lane
# streaming reads as 64 B; the factor is not assumed but CALIBRATED IN THE SAME PASS on a kernel whose read volume is
# known by construction (xHipMemCeilingDev's copy of the same buffer: it reads exactly n * 2048 bytes), so the headline
# kernel's figure does not rest on its own algorithmic byte count.
# ------------------------------------------------------------------------------------------------
def traffic_child(args):
    """what the profiled passes run: the calibration copy and the headline launch on the full batch, a few times, nothing else"""
    import x266_amd
    codec = x266_amd.Codec(0)
    n = args.dct_blocks
    x, z = codec.alloc(n * 2048), codec.alloc(n * 2048)
    codec.fill_residual_dev(x.ptr, n * 1024, DCT_SEED, 0, 0)
    for _ in range(3):
        codec.mem_ceiling_dev(0, x.ptr, z.ptr, n * 2048, 0)
    for _ in range(6):
        codec.dct32_fwd_dev(x.ptr, z.ptr, n, 0)
    codec.stream_sync()


def measure_traffic_live(n_dct, budget_s=150.0):
    """(bytes per headline launch, how it was measured) or (None, why not)"""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    # already running under a profiler (someone profiles this bench run): do not nest a second one
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself being profiled (ROCPROF* / ROCP_* in the environment)"
    t0 = time.time()
    kb, cal = {}, {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="x266_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            left = budget_s - (time.time() - t0)
            if left < 20:
                return None, "time budget of the live traffic passes exhausted"
            subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                            os.path.abspath(__file__), "--traffic-child", "--dct-blocks", str(n_dct)],
                           cwd="/tmp", env=env, timeout=left, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            vals, cvals = [], []
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r.get("Kernel_Name", "").replace(" ", "")
                    if r.get("Counter_Name") != ctr:
                        continue
                    if "dct32_lds_kernel<false>" in name or "dct32_lds_kernel<0" in name:
                        vals.append(float(r["Counter_Value"]))
                    elif "mem_ceiling_kernel<0" in name:
                        cvals.append(float(r["Counter_Value"]))
            if not vals:
                return None, "no %s rows for the forward kernel in rocprofv3's output" % ctr
            kb[ctr] = sum(vals) / len(vals)
            cal[ctr] = (sum(cvals) / len(cvals)) if cvals else None
        except Exception as e:                                            # a profiler that cannot run must not cost the bench line
            return None, "rocprofv3 --pmc %s failed: %s" % (ctr, str(e)[:120])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    known = n_dct * 2048.0                                                # what the calibration copy reads and writes, by construction
    f_fetch = known / (cal["FETCH_SIZE"] * 1024.0) if cal["FETCH_SIZE"] else 2.0
    f_write = known / (cal["WRITE_SIZE"] * 1024.0) if cal["WRITE_SIZE"] else 1.0
    total = kb["FETCH_SIZE"] * 1024.0 * f_fetch + kb["WRITE_SIZE"] * 1024.0 * f_write
    return total, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around 6 launches of the headline kernel on "
                   "this batch; KB x 1024 x a factor calibrated in the same pass on xHipMemCeilingDev's copy of the same buffer (known to read and write %d bytes): "
                   "FETCH_SIZE x %.4f%s, WRITE_SIZE x %.4f%s; raw: headline FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB per launch; %.0f s"
                   % (int(known), f_fetch, "" if cal["FETCH_SIZE"] else " (calibration rows missing: the guide's gfx950 factor)", f_write,
                      "" if cal["WRITE_SIZE"] else " (calibration rows missing)", kb["FETCH_SIZE"], kb["WRITE_SIZE"], time.time() - t0))


# ------------------------------------------------------------------------------------------------
# CPU baseline (checker leg: the only place bench.py touches oracle/)
# ------------------------------------------------------------------------------------------------
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


# ------------------------------------------------------------------------------------------------
# sysfs facts of the device (matched by PCI address: a box shows every GPU of the host, not only its own)
# ------------------------------------------------------------------------------------------------
def gpu_sysfs_dir(pci):
    import glob
    if not pci:
        return None
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.realpath(dev).split("/")[-1].lower() == pci:
            return dev
    return None


class SclkSampler:
    """median shader clock (MHz) of the device while a leg runs, read from its hwmon freq1_input every 10 ms by a thread;
    None where sysfs does not show it.  The VALU floors of the motion searches scale with it."""

    def __init__(self, sysfs_dir):
        import glob
        files = glob.glob(sysfs_dir + "/hwmon/hwmon*/freq1_input") if sysfs_dir else []
        self.path = files[0] if files else None
        self.samples = []

    def __enter__(self):
        self.stop = False
        if self.path:
            def run():
                while not self.stop:
                    try:
                        self.samples.append(int(open(self.path).read()) / 1e6)
                    except (OSError, ValueError):
                        pass
                    time.sleep(0.01)
            self.thread = threading.Thread(target=run, daemon=True)
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop = True
        if self.path:
            self.thread.join()

    def mhz(self):
        busy = [v for v in self.samples if v > 600]                      # idle samples between launches are not the kernel's clock
        return statistics.median(busy) if busy else None


def pcie_link_facts(sysfs_dir):
    """negotiated generation / width of the GPU's PCIe link from sysfs"""
    if not sysfs_dir:
        return None
    try:
        return {"speed": open(sysfs_dir + "/current_link_speed").read().strip(), "width": open(sysfs_dir + "/current_link_width").read().strip(),
                "device": os.path.realpath(sysfs_dir).split("/")[-1]}
    except OSError:
        return None


def trimmed_mean(d, frac=0.10):
    """mean of the middle 1 - 2 frac of the sorted durations"""
    s = sorted(d)
    k = int(len(s) * frac)
    s = s[k:len(s) - k] if len(s) - 2 * k >= 1 else s
    return sum(s) / len(s)


def smooth_frame_pair(w, h, pad, seed, mv=(5, -3)):
    """(cur [h, w], padded reference [h + 2 pad, w + 2 pad]) uint8: low-passed noise, the reference displaced by `mv` (planted motion)"""
    rs = np.random.RandomState(seed)
    big = rs.randint(0, 256, (h + 2 * pad + 16, w + 2 * pad + 16)).astype(np.float32)
    c = np.cumsum(np.pad(big, ((3, 2), (3, 2)), mode="edge"), axis=0)
    c = c[5:] - c[:-5]
    c = np.cumsum(c, axis=1)
    sm = (c[:, 5:] - c[:, :-5]) / 25.0                                   # 5x5 box low-pass so that motion is findable
    sm = np.clip((sm - 128.0) * 3.0 + 128.0, 0, 255).astype(np.uint8)
    cur = np.ascontiguousarray(sm[pad + 8:pad + 8 + h, pad + 8:pad + 8 + w])
    refp = np.ascontiguousarray(sm[8 - mv[1]:8 - mv[1] + h + 2 * pad, 8 - mv[0]:8 - mv[0] + w + 2 * pad])
    return cur, refp


# ------------------------------------------------------------------------------------------------
# the run: device, ranks, timing helpers
# ------------------------------------------------------------------------------------------------
def pick_device(local_rank, n_dev, share, env):
    """HIP device ordinal of the rank torchrun calls LOCAL_RANK: one process per GPU, rank i on device i.  A launcher that instead narrows
    every rank to ONE visible device (HIP_/ROCR_/CUDA_VISIBLE_DEVICES set per process) leaves ordinal 0 as the rank's own.  Anything
    else with fewer devices than ranks is a launch error, said here rather than as a failed hipSetDevice."""
    if share:
        return local_rank % n_dev
    if local_rank < n_dev:
        return local_rank
    if n_dev == 1 and any(env.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
        return 0
    raise SystemExit("LOCAL_RANK %d but only %d HIP device(s) visible: bench.py runs one rank per GPU" % (local_rank, n_dev))


class Bench:
    def __init__(self, args):
        import x266_amd                                                 # the library first: ITS HIP runtime is the process's
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch one rank per GPU (or run without torch.distributed.run: bench.py spawns its own ranks)" % (args.gpus, self.world))
        n_dev = int(x266_amd.load_library().xHipDeviceCount())
        if n_dev < 1:
            raise SystemExit("bench.py needs an MI355X: libx266hip has no CPU path")
        # Test hook (never set by the driver): X266_BENCH_SHARE_GPU=1 lets several ranks share the visible GPUs, so that the N > 1 code
        # path -- shard offsets, max-over-ranks timing, checksum reduction -- can be exercised on a one-GPU box.  RCCL refuses two ranks
        # on one device, so the node-layer legs run there only when X266HIP_RCCL_LIB names the tests' RCCL model.
        self.share = os.environ.get("X266_BENCH_SHARE_GPU") == "1"
        self.local_rank = pick_device(local_rank, n_dev, self.share, os.environ)
        self.codec = x266_amd.Codec(self.local_rank)
        self.hip = HipRuntime()
        self.hip.check(self.hip.lib.hipSetDevice(self.local_rank), "hipSetDevice")
        self.dist = None
        if self.world > 1:                                              # control plane only, on gloo, AFTER the library (module docstring)
            import torch.distributed as dist
            import torch
            self.torch = torch
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            self.dist = dist
        self.K, self.W = args.steps, args.warmup
        self.n_dct, self.n_satd = args.dct_blocks, args.satd_blocks
        self.stream = 0                                                 # every launch and event of the legs: the NULL stream
        self.events = [self.codec.event_create() for _ in range(max(self.K, 64) + 1)]
        self.info = self.codec.device_info()
        self.sysfs = gpu_sysfs_dir(self.hip.pci_bus_id(self.local_rank))
        self.ceil = {}                                                  # this box's streams, bytes per second (leg_same_box)
        self._keep = []

    # -- device memory through the C ABI -----------------------------------------------------------
    def dev(self, nbytes):
        return self.codec.alloc(max(int(nbytes), 16))

    def dev_from(self, arr):
        arr = np.ascontiguousarray(arr)
        b = self.dev(arr.nbytes)
        b.upload(arr)
        return b

    def dev_random_bytes(self, nbytes, seed):
        """uniform low bytes, sign-extension high bytes (the residual stream seen as bytes): contents never steer a kernel here"""
        b = self.dev(nbytes)
        self.codec.fill_residual_dev(b.ptr, nbytes // 2, seed, 0, self.stream)
        return b

    # -- control plane ------------------------------------------------------------------------------
    def barrier(self):
        self.hip.device_sync()
        if self.dist is not None:
            self.dist.barrier()
            self.hip.device_sync()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(self, value):
        """the value of every rank, in rank order (control-plane all-gather)"""
        if self.dist is None:
            return [value]
        t = self.torch.zeros(self.world, dtype=self.torch.float64)
        t[self.rank] = value
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def sum_over_ranks_i64(self, value):
        if self.dist is None:
            return int(value)
        t = self.torch.tensor([int(value)], dtype=self.torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    # -- timing ---------------------------------------------------------------------------------------
    def timed_leg(self, fn, steps=None, warmup=None):
        """fn() enqueues one step on the NULL stream.  Returns dict(wall_s, ms_per_step, kernel_ms (trimmed mean), kernel_ms_mean, kernel_ms_median)."""
        codec, events = self.codec, self.events
        steps = self.K if steps is None else max(1, min(steps, len(events) - 1))
        warmup = self.W if warmup is None else warmup
        self.hip.device_sync()                                          # clock pre-warm: a few launches to size one step, then ~PREWARM_SECONDS of load
        t0 = time.perf_counter()
        for _ in range(2):
            fn()
        self.hip.device_sync()
        per = max((time.perf_counter() - t0) / 2, 1e-6)
        pre = min(2000, int(PREWARM_SECONDS / per))
        for _ in range(pre + warmup):
            fn()
        self.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            codec.event_record(events[i], self.stream)
            fn()
        codec.event_record(events[steps], self.stream)
        self.barrier()
        wall = self.max_over_ranks(time.perf_counter() - t0)
        d = [codec.event_elapsed_ms(events[i], events[i + 1]) for i in range(steps)]
        # kernel time like wall time: the slowest rank's (every rank runs the same launches on its own shard)
        trim_all = self.all_ranks(trimmed_mean(d))
        return {"wall_s": wall, "steps": steps, "ms_per_step": wall / steps * 1e3, "kernel_ms": max(trim_all),
                "kernel_ms_mean": max(self.all_ranks(sum(d) / steps)), "kernel_ms_median": max(self.all_ranks(statistics.median(d))),
                "kernel_ms_by_rank": trim_all if self.world > 1 else None, "clock_prewarm_launches": pre}

    def rate(self, leg, units_per_step):
        return self.world * units_per_step * leg["steps"] / leg["wall_s"]

    def of_box(self, achieved_bytes_per_s, kind):
        """fraction of this box's own stream of that kind"""
        return achieved_bytes_per_s / self.ceil[kind]

    def hbm(self, leg, bytes_per_step, box_kind="copy"):
        """{hbm_frac, frac_of_same_box_<kind>} of a leg, from the trimmed mean of its launches"""
        achieved = bytes_per_step / (leg["kernel_ms"] * 1e-3)
        return {"hbm_frac": achieved / HBM_PEAK_BYTES_PER_S, "frac_of_same_box_%s" % box_kind: self.of_box(achieved, box_kind)}

    def roofline(self, leg, bytes_per_unit, n_units, traffic=None, traffic_source=None, box_kind="copy"):
        per_launch = bytes_per_unit * n_units
        achieved = per_launch / (leg["kernel_ms"] * 1e-3)
        return {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_BYTES_PER_S, "frac_of_same_box_%s" % box_kind: self.of_box(achieved, box_kind),
                "traffic": traffic, "traffic_source": traffic_source,
                "kernel_ms_per_launch": leg["kernel_ms"], "kernel_ms_mean": leg["kernel_ms_mean"], "kernel_ms_median": leg["kernel_ms_median"],
                "kernel_ms_per_launch_is": "10 % trimmed mean of the HIP-event durations of the timed launches (events on the launching stream); with several ranks the slowest rank's",
                "frac_by_rank": [per_launch / (m * 1e-3) / HBM_PEAK_BYTES_PER_S for m in leg["kernel_ms_by_rank"]] if leg.get("kernel_ms_by_rank") else None,
                "frac_at_mean": per_launch / (leg["kernel_ms_mean"] * 1e-3) / HBM_PEAK_BYTES_PER_S,
                "frac_at_median": per_launch / (leg["kernel_ms_median"] * 1e-3) / HBM_PEAK_BYTES_PER_S,
                "algorithmic_bytes_per_launch": per_launch}

    @staticmethod
    def brief(leg):
        return {k: leg[k] for k in ("ms_per_step", "kernel_ms", "kernel_ms_mean", "kernel_ms_median")}

    def same_on_device(self, a, b, nbytes, chunk=256 << 20):
        """two device buffers hold the same bytes (downloaded in chunks and compared on the host)"""
        for off in range(0, nbytes, chunk):
            n = min(chunk, nbytes - off)
            ha, hb = np.empty(n, np.uint8), np.empty(n, np.uint8)
            self.codec._check(self.codec.L.xHipMemcpyD2H(self.codec.ctx, ha.ctypes.data, a.ptr + off, n), "xHipMemcpyD2H")
            self.codec._check(self.codec.L.xHipMemcpyD2H(self.codec.ctx, hb.ctypes.data, b.ptr + off, n), "xHipMemcpyD2H")
            if not np.array_equal(ha, hb):
                return False
        return True


# ------------------------------------------------------------------------------------------------
# the legs: each returns its sub-dict of the line
# ------------------------------------------------------------------------------------------------
def leg_same_box(b, x, z):
    """What THIS box's memory system gives the streaming launch shapes, with no arithmetic (xHipMemCeilingDev): the same-run reference
    every HBM-bound leg is also expressed in, because boxes of the pool differ by 3-10 % in what a plain stream reaches."""
    ceil_bytes = b.n_dct * 2048
    for kind, name, moved in ((0, "copy", 2 * ceil_bytes), (1, "read", ceil_bytes), (3, "read_probe", ceil_bytes), (2, "write", ceil_bytes)):
        leg = b.timed_leg(lambda k=kind: b.codec.mem_ceiling_dev(k, x.ptr, z.ptr, ceil_bytes, b.stream), steps=min(b.K, 40), warmup=min(b.W, 10))
        b.ceil[name] = moved / (leg["kernel_ms"] * 1e-3)
    return {"copy_TBps": b.ceil["copy"] / 1e12, "read_TBps": b.ceil["read"] / 1e12, "read_no_store_TBps": b.ceil["read_probe"] / 1e12,
            "write_TBps": b.ceil["write"] / 1e12,
            "how": "xHipMemCeilingDev on the headline input / output buffers (%d MiB), trimmed mean of the timed launches' HIP-event durations, slowest rank: "
                   "nontemporal 16 B/lane streams in the launch shape that measured fastest for each (copy = the transform kernels' pattern, read = one XOR "
                   "checksum per 2 KiB, read_no_store = the same loads with nothing flowing back, write = the intra predictor's pattern)" % (ceil_bytes >> 20)}


def traffic_of_headline(b):
    """(pmc dict, source text): HBM bytes per launch from the counters -- measured live at N = 1, else replayed from profiles/traffic.json"""
    pmc, pmc_src = {}, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and b.n_dct == DCT_BLOCKS_PER_GPU and b.n_satd == SATD_BLOCKS_PER_GPU:
        try:
            pmc = json.load(open(tpath))
            pmc_src = "replayed from %s (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command); not measured in this run" % pmc.get("_source", "profiles/traffic.json")
        except Exception:
            pmc = {}
    if b.world == 1 and not b.args.no_live_traffic:
        b.hip.device_sync()
        live, how = measure_traffic_live(b.n_dct)
        if live is not None:
            pmc = dict(pmc, dct32_fwd_bytes_per_launch=live)
            pmc_src = how
        elif pmc_src:
            pmc_src += " (live passes: %s)" % how
        else:
            pmc_src = "not measured: " + how
    return pmc, pmc_src


def leg_dct32_inverse_and_fused(b, x, z, pmc, pmc_src):
    """inverse DCT32 of the headline's coefficients, and coefficients + reconstruction from one pass (6144 B per block)"""
    n, codec, out = b.n_dct, b.codec, {}
    r = b.dev(n * 2048)
    leg = b.timed_leg(lambda: codec.dct32_inv_dev(z.ptr, r.ptr, n, b.stream))
    xs, rs = x.download(np.int16, 4096 * 1024).astype(np.int32), r.download(np.int16, 4096 * 1024).astype(np.int32)
    out["dct32_inv"] = dict(value=b.rate(leg, n), unit="blocks/s", **b.brief(leg),
                            roofline=b.roofline(leg, DCT_BYTES_PER_BLOCK, n, pmc.get("dct32_inv_bytes_per_launch"), pmc_src),
                            parity="unpinned (no inverse in the reference); bit-exact vs this repo's oracle",
                            roundtrip_max_abs_err=int(np.abs(rs - xs).max()))
    z2 = b.dev(n * 2048)
    leg = b.timed_leg(lambda: codec.dct32_fwd_inv_dev(x.ptr, z2.ptr, r.ptr, n, b.stream))
    out["dct32_fwd_inv_fused"] = dict(value=b.rate(leg, n), unit="blocks/s", **b.brief(leg), **b.hbm(leg, 6144.0 * n),
                                      same_bytes_as_two_kernels=b.same_on_device(z2, z, n * 2048),
                                      note="2 KiB in, 2 + 2 KiB out per block; inverse fed from the forward's registers, inputs by LDS-DMA (DESIGN.md 3.7)")
    leg = b.timed_leg(lambda: codec.dct32_fwd_inv_dev(x.ptr, 0, r.ptr, n, b.stream), steps=max(4, b.K // 4), warmup=3)
    out["dct32_fwd_inv_fused"]["reconstruction_only"] = dict(value=b.rate(leg, n), unit="blocks/s", **b.brief(leg), **b.hbm(leg, 4096.0 * n),
                                                             note="d_coef = NULL: 2 KiB in, 2 KiB out, the natural-orientation pass 2 is skipped")
    return out


def leg_autotuned(b, x, z):
    """The opt-in "autotune" option (include/x266hip.h): the families whose fastest launch shape differs from box to box, default shape and the shape
    this box's first large call kept, alternating on the same buffers.  The headline and every other leg of the line run the DEFAULTS."""
    codec, n = b.codec, b.n_dct
    r, z2 = b.dev(n * 2048), b.dev(n * 2048)
    ns = b.n_satd
    d, s = b.dev(ns * 128), b.dev(ns * 4)
    codec.fill_residual_dev(d.ptr, ns * 64, SATD_SEED, 0, b.stream)
    nsad = ns * 128 // 2 // 64
    legs = (("dct32_fwd_inv_fused", 6144.0 * n, lambda: codec.dct32_fwd_inv_dev(x.ptr, z2.ptr, r.ptr, n, b.stream)),
            ("dct32_reconstruction_only", 4096.0 * n, lambda: codec.dct32_fwd_inv_dev(x.ptr, 0, r.ptr, n, b.stream)),
            ("satd8x8", float(SATD_BYTES_PER_BLOCK) * ns, lambda: codec.satd8x8_dev(d.ptr, s.ptr, ns, b.stream)),
            ("sad_8x8", 132.0 * nsad, lambda: codec.sad_dev(8, d.ptr, d.ptr + ns * 64, s.ptr, nsad, b.stream)))
    out, short = {}, max(4, b.K // 4)
    for name, nbytes, fn in legs:
        fr = {}
        for mode in (0, 1, 0, 1):                                        # default, tuned, default, tuned: the better of two per mode
            codec.set_option("autotune", mode)
            leg = b.timed_leg(fn, steps=short, warmup=3)
            f = nbytes / (leg["kernel_ms"] * 1e-3) / HBM_PEAK_BYTES_PER_S
            fr[mode] = max(fr.get(mode, 0.0), f)
        out[name] = {"default_hbm_frac": fr[0], "autotuned_hbm_frac": fr[1]}
    codec.set_option("autotune", 0)
    rep = codec.autotune_report()
    names = {"dct32_fwd_inv_fused": "dct32_fwd_inv", "dct32_reconstruction_only": "dct32_recon_only", "satd8x8": "satd8x8", "sad_8x8": "sad8"}
    for name, fam in names.items():
        out[name].update(choice=rep.get(fam, {}).get("choice"), candidate_ms=rep.get(fam, {}).get("ms"))
    out["same_bytes_default_and_tuned"] = b.same_on_device(z2, z, n * 2048)
    out["note"] = "candidate 0 = the default shape; candidates in x266hip_abi.hip (kFwdInvCands, kReconCands, kSatdCands, kSadCands)"
    return out


def leg_satd(b, pmc, pmc_src):
    """the 8x8 SATD residual batch (the secondary metric), with the port timed on the host cores at N = 1"""
    n, codec = b.n_satd, b.codec
    d, s = b.dev(n * 128), b.dev(n * 4)
    codec.fill_residual_dev(d.ptr, n * 64, SATD_SEED, b.rank * n * 64, b.stream)
    leg = b.timed_leg(lambda: codec.satd8x8_dev(d.ptr, s.ptr, n, b.stream))
    out = dict(value=b.rate(leg, n), unit="blocks/s", blocks_per_gpu=n, **b.brief(leg),
               roofline=b.roofline(leg, SATD_BYTES_PER_BLOCK, n, pmc.get("satd8x8_bytes_per_launch"), pmc_src, box_kind="read"))
    if b.rank == 0 and b.world == 1 and not b.args.no_cpu_baseline:
        ns = min(n, 1 << 23)
        out["cpu_baseline"] = cpu_baseline_satd(d.download(np.int16, ns * 64).reshape(ns, 64), s.download(np.uint32, ns))
    return out


def leg_motion_search(b, keep):
    """BASELINE configs[2]: full-search motion estimation of one 3840x2160 luma frame, window +-64 -- SATD, then the SAD metric"""
    codec = b.codec
    w, h, rng = 3840, 2160, 64
    cur_h, refp_h = smooth_frame_pair(w, h, rng, 0x266 + b.rank)           # planted motion (5, -3)
    cur, refp = b.dev_from(cur_h), b.dev_from(refp_h)
    rstride = refp_h.shape[1]
    nb = (w // 8) * (h // 8)
    best = b.dev(nb * 8)
    origin = refp.ptr + rng * rstride + rng
    keep.update(cur=cur, refp=refp, best=best, origin=origin, rstride=rstride)
    ncand = nb * (2 * rng + 1) ** 2
    me_steps = max(4, b.K // 4)
    out = {}
    # VALU floors: 32 x v_sad_u16 (SATD) / 16 x v_sad_u8 (SAD) per 64 candidates, 4 cycles per wave64 instruction (tools/probes/alubench), at the
    # 2.4 GHz the part is specified for and -- where sysfs shows it -- at the shader clock this box sustained during the leg
    for key, fn, per64, unit, floor_name, steps in (("satd8x8_me_search", codec.satd_search_dev, 32, "SATD/s", "v_sad_u16", max(me_steps, 40)),
                                                    ("sad8x8_me_search", codec.sad_search_dev, 16, "SAD/s", "v_sad_u8", max(me_steps, 60))):
        with SclkSampler(b.sysfs) as clk:
            leg = b.timed_leg(lambda f=fn: f(cur.ptr, w, origin, rstride, w, h, rng, best.ptr, 0, b.stream), steps=steps, warmup=2)
        mv = best.download(np.int16, nb * 4).reshape(nb, 4)[:, :2]
        cycles = ncand / 64 * per64 * 4 / (4 * b.info["cu_count"])
        sclk = clk.mhz()
        out[key] = dict(value=b.rate(leg, ncand), unit=unit, ms_per_frame=leg["ms_per_step"], **b.brief(leg), sclk_mhz=sclk,
                        planted_mv_found_fraction=float(((mv[:, 0] == 5) & (mv[:, 1] == -3)).mean()))
        out[key]["frac_of_%s_floor" % floor_name] = cycles / 2.4e9 / (leg["kernel_ms"] * 1e-3)
        out[key]["frac_of_%s_floor_at_sclk" % floor_name] = (cycles / (sclk * 1e6) / (leg["kernel_ms"] * 1e-3)) if sclk else None
    out["satd8x8_me_search"].update(
        frame="%dx%d luma, 8x8 blocks, window +-%d (%d candidates per block)" % (w, h, rng, (2 * rng + 1) ** 2),
        bound="VALU issue (v_sad_u16), not HBM: ~18 MB of compulsory traffic per frame",
        parity="per-candidate cost pinned by satd8x8 (src_tb/satd.c); harness (order, tie-break, padding) unpinned")
    out["sad8x8_me_search"]["parity"] = "metric = sad() of riscv/programs/benchmarks/sad/sad.c at n = 8; harness unpinned, as for the SATD search"
    return out


def leg_transform_set(b, x):
    """BASELINE configs[3]: the mixed transform set (DCT-II 4..32 + closed-form DST-VII 4/8/16), 2 GiB of residual per class,
    and the CTU-ordered mixed buffer in one launch"""
    codec, n_dct = b.codec, b.n_dct
    zt = b.dev(n_dct * 2048)                                            # own output buffer: z still holds the headline leg's result
    short = max(4, b.K // 4)
    ts = {}
    for ttype, tname, inverse in ((0, "dct2", False), (1, "dst7", False), (0, "dct2_inv", True), (1, "dst7_inv", True)):
        for n in (4, 8, 16):
            nblk = (n_dct * 1024) // (n * n)
            call = codec.transform_inv_dev if inverse else codec.transform_fwd_dev
            leg = b.timed_leg(lambda c=call, tt=ttype, nn=n, cnt=nblk: c(tt, nn, x.ptr, zt.ptr, cnt, 0, b.stream), steps=short, warmup=3)
            ts["%s_%dx%d" % (tname, n, n)] = dict(value=b.rate(leg, nblk), unit="blocks/s", **b.hbm(leg, 4.0 * n * n * nblk), **b.brief(leg))
    # per-CTU mixed batch: every 64x64 CTU's 32x32 quadrants cycle through the seven (type, size) classes; every quadrant is a tile
    # with its own class byte, the whole buffer is ONE launch (xTransformTilesDev)
    n_ctu = (n_dct * 1024) // 4096
    q = np.arange(n_ctu * 4, dtype=np.int64)
    tile_cls = b.dev_from(np.array([3, 2, 6, 1, 5, 0, 4], np.uint8)[(q + q // 4) % 7])   # kinds -> type*4 + log2N-2
    per_ctu = {"layout": "64x64 CTUs whose 32x32 quadrants cycle through the seven classes (DCT-II 32/16/8/4, DST-VII 16/8/4), "
                         "TUs of a quadrant contiguous", "ctus": n_ctu}
    for inv_flag, name in ((0, "per_ctu_one_launch"), (1, "per_ctu_one_launch_inverse")):
        leg = b.timed_leg(lambda f=inv_flag: codec.transform_tiles_dev(f, x.ptr, zt.ptr, n_ctu * 4, 0, tile_cls.ptr, b.stream), steps=short, warmup=3)
        per_ctu[name] = dict(value=b.rate(leg, n_ctu), unit="CTUs/s", **b.hbm(leg, 4.0 * n_ctu * 4096), **b.brief(leg))
    return {"classes": ts, "per_ctu_mixed": per_ctu, "parity": "unpinned upstream except DCT-II 32; bit-exact vs this repo's oracle",
            "note": "4*N*N algorithmic bytes per block; hbm_frac from the trimmed mean of the timed launches' HIP-event durations; the seven-calls-over-offset-"
                    "tables form of the mixed buffer (rounds 1-4: 0.59 of 8 TB/s, superseded by the one-launch call) is still tested, no longer benched"}


def leg_fused_from_tiles(b):
    """Tiled cur / pred frames -> coefficients / costs, residual never in HBM, next to the two-kernel paths.  32768^2 luma: exactly 2^20
    DCT32 blocks and 2^24 SATD blocks, i.e. the two-kernel legs launch the headline kernels at the headline sizes."""
    codec = b.codec
    fw, fh = 32768, 32768
    ntile = (fw // 16) * (fh // 16)
    tcur, tpred = b.dev_random_bytes(ntile * 512, 0x266), b.dev_random_bytes(ntile * 512, 0x268)
    fcoef, fcost, fres = b.dev(fw * fh * 2), b.dev(fw * fh // 64 * 4), b.dev(fw * fh * 2)
    short = max(4, b.K // 4)
    fused = {}
    legs = (("dct32_from_tiles", fw * fh // 1024, 4096, lambda: codec.dct32_fwd_from_tiles_dev(tcur.ptr, tpred.ptr, fw, fh, fcoef.ptr, b.stream)),
            ("dct32_residual_then_transform", fw * fh // 1024, None,
             lambda: (codec.residual_luma_dev(tcur.ptr, tpred.ptr, fw, fh, 32, fres.ptr, b.stream), codec.dct32_fwd_dev(fres.ptr, fcoef.ptr, fw * fh // 1024, b.stream))),
            ("satd8x8_from_tiles", fw * fh // 64, 132, lambda: codec.satd8x8_from_tiles_dev(tcur.ptr, tpred.ptr, fw, fh, fcost.ptr, b.stream)),
            ("satd8x8_residual_then_cost", fw * fh // 64, None,
             lambda: (codec.residual_luma_dev(tcur.ptr, tpred.ptr, fw, fh, 8, fres.ptr, b.stream), codec.satd8x8_dev(fres.ptr, fcost.ptr, fw * fh // 64, b.stream))))
    # the chroma half (m_C of the same tiles, src/x266.cpp:60): per 64x64 CTU one 32x32 U and one 32x32 V block -> 2^19 DCT32 blocks; per tile
    # one 8x8 U and V block -> 2^23 SATD blocks.  Planar U / V output streams.  The read side touches ONE 128-byte line of every 512-byte tile.
    nc32, nc8 = fw * fh // 4096 * 2, ntile * 2
    legs += (("chroma_dct32_from_tiles", nc32, 4096,
              lambda: codec.dct32_fwd_chroma_from_tiles_dev(tcur.ptr, tpred.ptr, fw, fh, fcoef.ptr, fcoef.ptr + nc32 * 1024, 1, b.stream)),
             ("chroma_satd8x8_from_tiles", nc8, 132,
              lambda: codec.satd8x8_chroma_from_tiles_dev(tcur.ptr, tpred.ptr, fw, fh, fcost.ptr, fcost.ptr + ntile * 4, 1, b.stream)),
             ("residual_chroma_32", nc32, 4096,
              lambda: codec.residual_chroma_dev(tcur.ptr, tpred.ptr, fw, fh, 32, fres.ptr, fres.ptr + nc32 * 1024, 1, b.stream)),
             ("residual_chroma_8", nc8, 256,
              lambda: codec.residual_chroma_dev(tcur.ptr, tpred.ptr, fw, fh, 8, fres.ptr, fres.ptr + nc8 * 64, 1, b.stream)))
    for name, units, bytes_per_unit, fn in legs:
        leg = b.timed_leg(fn, steps=short, warmup=3)
        fused[name] = dict(value=b.rate(leg, units), unit="blocks/s", **b.brief(leg))
        if bytes_per_unit:
            fused[name].update(b.hbm(leg, bytes_per_unit * units, "read" if "satd" in name else "copy"))
    fused["note"] = ("%dx%d tiled frame pair (x266.cpp ref_block_t); fused kernels are bit-identical to the two-kernel paths "
                     "listed next to them (tests/test_gpu_tiles.py)" % (fw, fh))
    return fused


def leg_front_end_and_sad(b):
    """frame container conversion, residual formation, SAD -- pure data movement, HBM-bound"""
    codec = b.codec
    fw2, fh2 = 16384, 16384                                              # 256 Mi luma samples
    npx = fw2 * fh2
    ypl = b.dev_random_bytes(npx, 0x77 + b.rank)
    upl, vpl = b.dev_random_bytes(npx // 4, 0x78), b.dev_random_bytes(npx // 4, 0x79)
    t_a, t_b = b.dev_random_bytes(npx * 2, 0x7a), b.dev_random_bytes(npx * 2, 0x7b)   # 512-byte tiles: 2 bytes per luma sample
    res2, sad_o = b.dev(npx * 2), b.dev(npx // 64 * 4)
    short = max(4, b.K // 4)
    front = {}
    for name, nbytes, fn in (
            ("conv_input_fmt", 3.0 * npx, lambda: codec.conv_input_fmt_dev(t_a.ptr, ypl.ptr, upl.ptr, vpl.ptr, fw2, fw2, fh2, b.stream)),
            ("conv_output_420", 3.0 * npx, lambda: codec.conv_output_420_dev(t_a.ptr, ypl.ptr, fw2, upl.ptr, vpl.ptr, fw2 // 2, fw2, fh2, b.stream)),
            ("residual_luma_32", 4.0 * npx, lambda: codec.residual_luma_dev(t_a.ptr, t_b.ptr, fw2, fh2, 32, res2.ptr, b.stream)),
            ("sad_8x8", 2.0 * npx + 4.0 * (npx // 64), lambda: codec.sad_dev(8, ypl.ptr, t_b.ptr, sad_o.ptr, npx // 64, b.stream)),
            ("sad_16x16", 2.0 * npx + 4.0 * (npx // 256), lambda: codec.sad_dev(16, ypl.ptr, t_b.ptr, sad_o.ptr, npx // 256, b.stream)),
            ("sad_64x64", 2.0 * npx + 4.0 * (npx // 4096), lambda: codec.sad_dev(64, ypl.ptr, t_b.ptr, sad_o.ptr, npx // 4096, b.stream))):
        leg = b.timed_leg(fn, steps=short, warmup=3)
        front[name] = dict(GBps=b.world * nbytes * leg["steps"] / leg["wall_s"] / 1e9, **b.hbm(leg, nbytes, "read" if name.startswith("sad") else "copy"),
                           samples_per_s=b.world * npx * leg["steps"] / leg["wall_s"], **b.brief(leg))
    front["note"] = ("%dx%d frame; bytes = planes read + tile bytes written (conv), luma of both tile frames + int16 residual "
                     "(residual), both blocks + 4-byte result (sad)" % (fw2, fh2))
    return front


def leg_intra(b):
    """32x32 intra prediction, mode decision, and prediction -> residual -> DCT32 in one kernel (HEVC 35 modes; parity unpinned upstream)"""
    codec = b.codec
    n_sets = 59918                                                       # x 35 modes = 2 GiB of predictions
    rs = np.random.RandomState(0x32 + b.rank)
    refs_t = b.dev_from(rs.randint(0, 256, (n_sets, 144)).astype(np.uint8))
    modes_t = b.dev_from(np.tile(np.arange(35, dtype=np.uint8), n_sets))
    index_t = b.dev_from(np.repeat(np.arange(n_sets, dtype=np.int32), 35))
    pred_t = b.dev(n_sets * 35 * 1024)
    n_dec = min(1 << 17, n_sets)
    src_t = b.dev_random_bytes(n_dec * 1024, 0x33)
    cost_t, bestm_t = b.dev(n_dec * 35 * 4), b.dev(n_dec)
    short = max(4, b.K // 4)
    intra = {}
    leg = b.timed_leg(lambda: codec.intra32_predict_dev(refs_t.ptr, modes_t.ptr, index_t.ptr, pred_t.ptr, n_sets * 35, b.stream), steps=short, warmup=3)
    written = 1024.0 * n_sets * 35 / (leg["kernel_ms"] * 1e-3)
    intra["predict"] = dict(value=b.rate(leg, n_sets * 35), unit="predictions/s", written_hbm_frac=written / HBM_PEAK_BYTES_PER_S,
                            frac_of_same_box_write=b.of_box(written, "write"), **b.brief(leg))
    leg = b.timed_leg(lambda: codec.intra32_costs_dev(refs_t.ptr, src_t.ptr, cost_t.ptr, bestm_t.ptr, n_dec, b.stream), steps=short, warmup=3)
    intra["decide_35_modes"] = dict(value=b.rate(leg, n_dec), unit="blocks/s", satd8x8_per_s=b.rate(leg, n_dec) * 35 * 16, **b.brief(leg))
    if hasattr(codec, "intra32_residual_dct32_dev"):
        # the encoder loop's form: the chosen mode's prediction never reaches HBM (1 KiB of source + 144 B of references in, 2 KiB of coefficients out)
        n_blk = min(1 << 20, n_sets * 35)
        src_b, coef_b = b.dev_random_bytes(n_blk * 1024, 0x34), b.dev(n_blk * 2048)
        leg = b.timed_leg(lambda: codec.intra32_residual_dct32_dev(refs_t.ptr, modes_t.ptr, index_t.ptr, src_b.ptr, coef_b.ptr, n_blk, b.stream), steps=short, warmup=3)
        intra["predict_residual_dct32"] = dict(value=b.rate(leg, n_blk), unit="blocks/s", **b.hbm(leg, 3072.0 * n_blk), **b.brief(leg),
                                               note="xIntra32ResidualDct32Dev: predict (given mode) -> src - pred -> forward DCT32 in one kernel; 1 KiB in + 2 KiB out per block")
    intra["parity"] = "unpinned upstream (src/mkIntra32-wip.bsv is a sketch without a model); bit-exact vs this repo's oracle"
    return intra


def leg_host_api(b, n):
    """The literal drop-in path: xDct32FwdBatch on n blocks from pageable and from pinned host buffers (what INTEGRATION.md section 2 tells an
    x266.cpp maintainer to call, src/x266.cpp:526-555), next to what the link itself gives (plain copies).  PCIe-inclusive -- never `value`."""
    codec, hip = b.codec, b.hip
    nbytes = n * 2048
    out = {"blocks": n, "MiB_each_way": nbytes >> 20, "pcie_link": pcie_link_facts(b.sysfs)}

    def best_of(fn, reps=4):
        best = 1e9
        for _ in range(reps):
            hip.device_sync()
            t0 = time.perf_counter()
            fn()
            hip.device_sync()
            best = min(best, time.perf_counter() - t0)
        return best
    hp_in, hp_out = codec.host_alloc(nbytes, np.uint8), codec.host_alloc(nbytes, np.uint8)
    d_a, d_b = b.dev(nbytes), b.dev(nbytes)
    streams = [hip.stream_create() for _ in range(4)]
    H2D, D2H = 1, 2                                                      # hipMemcpyHostToDevice / DeviceToHost

    def both(s1, s2):
        hip.memcpy_async(d_a.ptr, hp_in.ctypes.data, nbytes, H2D, s1)
        hip.memcpy_async(hp_out.ctypes.data, d_b.ptr, nbytes, D2H, s2)
    # HIP multiplexes streams onto a few hardware queues; two streams that land on the same one serialise their copies. The link's
    # two-way rate is what the best of a few stream pairs reaches.
    two_way = max(nbytes / best_of(lambda a=a, c=c: both(streams[a], streams[c]), reps=2) / 1e9 for a, c in ((0, 1), (0, 2), (1, 3), (2, 3)))
    out["link_GBps"] = {"h2d_alone": nbytes / best_of(lambda: hip.memcpy_async(d_a.ptr, hp_in.ctypes.data, nbytes, H2D, streams[0])) / 1e9,
                        "d2h_alone": nbytes / best_of(lambda: hip.memcpy_async(hp_out.ctypes.data, d_b.ptr, nbytes, D2H, streams[1])) / 1e9,
                        "each_way_both_directions_at_once": two_way,
                        "how": "one plain %d MiB hipMemcpyAsync from / to pinned memory per direction; two-way: best of four stream pairs" % (nbytes >> 20)}
    for s in streams:
        hip.lib.hipStreamDestroy(s)
    del hp_in, hp_out, d_a, d_b
    x_dev = b.dev(nbytes)
    codec.fill_residual_dev(x_dev.ptr, n * 1024, DCT_SEED, 0, 0)
    xh = x_dev.download(np.int16, n * 1024).reshape(n, 1024)            # pageable, touched
    zh = np.ones_like(xh)

    def call(i, o):
        rc = codec.L.xDct32FwdBatch(codec.ctx, P(i), P(o), n)
        if rc:
            raise RuntimeError("xDct32FwdBatch failed: %d" % rc)
    dt = best_of(lambda: call(xh.ctypes.data, zh.ctypes.data))
    out["pageable"] = {"blocks_per_s": n / dt, "GBps_each_way": nbytes / dt / 1e9, "ms": dt * 1e3}
    xp, zp = codec.host_alloc((n, 1024), np.int16), codec.host_alloc((n, 1024), np.int16)
    xp[:] = xh
    dt = best_of(lambda: call(xp.ctypes.data, zp.ctypes.data))
    out["pinned"] = {"blocks_per_s": n / dt, "GBps_each_way": nbytes / dt / 1e9, "ms": dt * 1e3, "same_result_as_pageable": bool(np.array_equal(zp, zh))}
    for k in ("pinned", "pageable"):
        out[k]["frac_of_link_both_directions"] = out[k]["GBps_each_way"] / two_way
    out["note"] = ("host pointers in and out, best of 4 calls: 16 MiB chunks over three staging slots, uploads + kernels issued by the calling thread, "
                   "downloads by a helper thread (a pageable copy blocks its issuing thread); inputs are NOT resident, so this is never `value`")
    return out


def leg_node_stream8k(b, node, also):
    """BASELINE configs[4]: the 7680x4320 frame stream through the node layer (one RCCL group per step with N > 1; in place with one rank)"""
    codec, rank, world = b.codec, b.rank, b.world
    fw8, fh8 = 7680, 4320
    nd8, ns8 = (fw8 // 32) * (fh8 // 32), (fw8 // 8) * (fh8 // 8)
    IN_RING, OUT_RING = 4, 5                                             # X266_STREAM_IN_RING / X266_STREAM_OUT_RING (include/x266hip.h)
    fin = fout = None
    if rank == 0:
        fin = [(b.dev(nd8 * 2048), b.dev(ns8 * 128)) for _ in range(IN_RING)]
        fout = [(b.dev(nd8 * 2048), b.dev(ns8 * 4)) for _ in range(OUT_RING)]
        for i, (a, d) in enumerate(fin):
            codec.fill_residual_dev(a.ptr, nd8 * 1024, DCT_SEED, i * 100000007, b.stream)
            codec.fill_residual_dev(d.ptr, ns8 * 64, SATD_SEED, i * 100000007, b.stream)
    b.hip.device_sync()
    st8 = node.frame_stream(fw8, fh8)
    # one foreign call per frame: the argument arrays of every (input ring, output ring) pairing are built once
    prep8 = ([st8.prepare([fin[i % IN_RING][0].ptr, fin[i % IN_RING][1].ptr], [fout[i % OUT_RING][0].ptr, fout[i % OUT_RING][1].ptr])
              for i in range(IN_RING * OUT_RING)] if rank == 0 else None)
    raw_next = node.L.xNodeStreamNextSlotStream

    def push8(f):
        if rank == 0:                                                    # resident inputs: "produced" on the frame's own slot stream, so the push needs no producer event
            st8.push_prepared(prep8[f % (IN_RING * OUT_RING)], raw_next(st8.s))
        else:
            st8.push()
    F = b.args.stream8k
    # clocks: ~0.1 s of frames before the timed ones (200 frames are 7 ms); a fixed count, the same on every rank
    for f in range(2500 if world == 1 else 64):
        push8(f)
    st8.flush()
    b.barrier()
    t0 = time.perf_counter()
    for f in range(F):
        push8(f)
    st8.flush()
    b.barrier()
    wall8 = b.max_over_ranks(time.perf_counter() - t0)
    exact8 = kernel_us = None
    if rank == 0:                                                        # last frame against the plain single-device calls
        a, d = fin[(F - 1) % IN_RING]
        c, e = fout[(F - 1) % OUT_RING]
        c1, e1 = b.dev(nd8 * 2048), b.dev(ns8 * 4)
        codec.dct32_fwd_dev(a.ptr, c1.ptr, nd8, b.stream)
        codec.satd8x8_dev(d.ptr, e1.ptr, ns8, b.stream)
        b.hip.device_sync()
        exact8 = b.same_on_device(c, c1, nd8 * 2048) and b.same_on_device(e, e1, ns8 * 4)
    if rank == 0 and world == 1:                                         # what the frame's one launch costs by itself, back to back on one stream
        a, d = fin[0]
        c, e = fout[0]
        for phase in (0, 1):
            for _ in range(200):
                codec.frame_lanes_dev(a.ptr, c.ptr, nd8, d.ptr, e.ptr, ns8, b.stream)
            codec.event_record(b.events[phase], b.stream)
        kernel_us = codec.event_elapsed_ms(b.events[0], b.events[1]) / 200 * 1e3
    link_bytes = (nd8 * 2048 + ns8 * 128) / world                        # one peer's input shard of a frame, over one link
    also["stream8k"] = {
        "frames_per_s": F / wall8, "ms_per_frame": wall8 / F * 1e3, "frames": F,
        "kernel_us": kernel_us, "launches_per_frame_and_rank": 1,
        "kernel_share_of_frame_time": (kernel_us * 1e-6 / (wall8 / F)) if kernel_us else None,
        "dct32_blocks_per_s": nd8 * F / wall8, "satd8x8_blocks_per_s": ns8 * F / wall8,
        "frame": "7680x4320: %d DCT32 blocks (66.4 MB) + %d SATD blocks (66.4 MB)" % (nd8, ns8),
        "path": "C ABI node layer (xNodeStreamPush / Flush): per step one RCCL group carries frame t's shards root -> peers and "
                "frame t-2's coefficients and costs peers -> root on a communication stream while every rank transforms frame t"
                if world > 1 else "C ABI node layer, one rank: the root transforms the frame in place, no transfer (RCCL only in the self-test)",
        "bit_exact_vs_single_device": exact8,
        "link_bound_frames_per_s": (XGMI_LINK_BYTES_PER_S / link_bytes) if world > 1 else None,
        "link_bound": "each peer's input shard crosses ONE xGMI link (~153 GB/s per direction): <= 7.5e7 DCT32 blocks/s per peer (SURVEY.md 8e)"}
    st8.close()


def leg_node_batch_and_search(b, node, also, x, z, me):
    """one resident batch scattered and gathered (SURVEY 8e "end-to-end scatter -> compute -> gather"), and the sharded motion search"""
    from x266_amd.node import OP_DCT32_FWD
    rank, world = b.rank, b.world
    nsg = min(1 << 18, b.n_dct)
    pin, pout = (x.ptr, z.ptr) if rank == 0 else (0, 0)
    b.hip.device_sync()
    node.batch_scatter_gather(OP_DCT32_FWD, pin, pout, nsg, 0)
    b.barrier()
    t0 = time.perf_counter()
    for _ in range(4):
        node.batch_scatter_gather(OP_DCT32_FWD, pin, pout, nsg, 0)
    b.barrier()
    wall_sg = b.max_over_ranks(time.perf_counter() - t0) / 4
    also["dct32_scatter_gather"] = {"value": nsg / wall_sg, "unit": "blocks/s", "blocks": nsg,
                                    "link_bound_blocks_per_s": (world * XGMI_LINK_BYTES_PER_S / 2048.0) if world > 1 else None,
                                    "link_bound": "every peer's shard crosses ONE xGMI link (~153 GB/s per direction, inputs one way, results the other): "
                                                  "<= 153e9 / 2048 = 7.5e7 blocks/s per peer, i.e. world x 7.5e7 with the root computing its own shard in place"
                                                  if world > 1 else None,
                                    "note": "root-resident batch cut into chunks (8 MiB of input per rank), pipelined through the node stream "
                                            "(xNodeBatchScatterGather); at N = 1 no transfer"}
    if not me:
        return
    # sharded motion search: stripes + halo from the root, records back (xNodeSatd8x8Search)
    nstr = max(world, 1)
    cur, origin, best = (me["cur"].ptr, me["origin"], me["best"].ptr) if rank == 0 else (0, 0, 0)
    nb8 = (3840 // 8) * (2160 // 8) * 8

    def search():
        node.satd_search(cur, 3840, origin, me["rstride"], 3840, 2160, 64, nstr, best)
    search()
    ref_best = me["best"].download(np.uint8, nb8) if rank == 0 else None
    b.barrier()
    t0 = time.perf_counter()
    for _ in range(3):
        search()
    b.barrier()
    wall_ms = b.max_over_ranks(time.perf_counter() - t0) / 3
    same = None
    if rank == 0:
        b.codec.satd_search_dev(cur, 3840, origin, me["rstride"], 3840, 2160, 64, best, 0, b.stream)
        same = bool(np.array_equal(me["best"].download(np.uint8, nb8), ref_best))
    also["satd8x8_me_search_sharded"] = {"ms_per_frame": wall_ms * 1e3, "stripes": nstr, "identical_to_single_device": same,
                                         "note": "synchronous call incl. scatter of cur stripes + reference halo and gather of (mv, cost)"}


def node_legs(b, also, x, z, me):
    """The node layer of the C ABI: BASELINE configs[4] and the other end-to-end scatter / gather figures -- the only legs that talk RCCL."""
    from x266_amd.node import Node
    if os.environ.get("X266_BENCH_TEST_STALL_RANK") == str(b.rank):         # test hook (never set by the driver): this rank never joins the node layer
        time.sleep(1e6)
    uid = [Node.unique_id() if b.rank == 0 else None]
    if b.dist is not None:
        b.dist.broadcast_object_list(uid, src=0)
    node = Node.for_rank(b.local_rank, b.rank, b.world, uid[0])      # xHipNodeInitRank: one process per GPU, also at N = 1
    node.self_test()                                                    # RCCL ring send/recv + all-reduce, checked
    ver, path = Node.rccl_info()
    infos = ["%s (version %d)" % (path, ver)]
    if b.dist is not None:
        infos = [None] * b.world
        b.dist.all_gather_object(infos, "%s (version %d)" % (path, ver))
    also["rccl_by_rank"] = infos                                        # which library each rank's node layer talks to
    leg_node_stream8k(b, node, also)
    leg_node_batch_and_search(b, node, also, x, z, me)
    node.close()


def run_node_legs_under_watchdog(b, result, also, x, z, me, json_fd):
    """A communication hang must not cost the whole line: past --node-timeout seconds rank 0 prints the JSON with what it has
    (the legs marked as timed out) and every rank leaves."""
    def node_timed_out():
        also["node_layer_error"] = "node-layer legs did not finish within %.0f s (RCCL hang?); line printed without them" % b.args.node_timeout
        if b.rank == 0:
            result.setdefault("cpu_baseline", None)
            result["device"] = b.info["name"].strip()
            emit(result, json_fd)
        os._exit(0)
    watchdog = threading.Timer(b.args.node_timeout, node_timed_out)
    watchdog.daemon = True
    watchdog.start()
    try:
        node_legs(b, also, x, z, me)
    except Exception as e:                                              # e.g. RCCL missing: keep the rest of the line
        also["node_layer_error"] = "%s: %s" % (type(e).__name__, e)
    watchdog.cancel()


# ------------------------------------------------------------------------------------------------
# the line the driver parses: compact (<= 4 KB), numbers only; the full record goes to bench_full.json
# ------------------------------------------------------------------------------------------------
COMPACT_LIMIT_BYTES = 4096             # target; tests/test_bench_line.py fails the build at 8192
FULL_RECORD = "bench_full.json"
_FRAC_KEYS = ("frac", "hbm_frac", "written_hbm_frac", "frac_of_v_sad_u16_floor", "frac_of_v_sad_u8_floor", "autotuned_hbm_frac")
_RATE_KEYS = ("value", "frames_per_s", "blocks_per_s", "ms_per_frame")
_GROUPS = ("transform_set", "classes", "per_ctu_mixed", "fused_from_tiles", "front_end_and_sad")   # containers, not legs: children keep their own names


def _sig(v, digits=5):
    """numbers of the compact line: 5 significant digits (ints and bools untouched)"""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    return float("%.*g" % (digits, v))


def _leg_scalars(name, leg, out, failed):
    """one leg of `also` -> {name: roofline fraction (else its rate)}; every boolean check of the leg must be true"""
    if not isinstance(leg, dict):
        return
    roof = leg.get("roofline") if isinstance(leg.get("roofline"), dict) else {}
    for k in _FRAC_KEYS:
        if isinstance(roof.get(k, leg.get(k)), (int, float)):
            out[name] = _sig(float(roof.get(k, leg.get(k))), 4)
            break
    else:
        for k in _RATE_KEYS:
            if isinstance(leg.get(k), (int, float)) and not isinstance(leg.get(k), bool):
                out[name] = _sig(float(leg[k]), 4)
                break
    for k, v in leg.items():
        if isinstance(v, bool) and k != "torch_imported" and not v:
            failed.append("%s.%s" % (name, k))
        elif isinstance(v, dict) and k not in ("roofline", "cpu_baseline", "pcie_link", "link_GBps"):
            _leg_scalars(k if name in _GROUPS else "%s.%s" % (name, k), v, out, failed)


def compact_record(full):
    """The single stdout line: the contract's top-level keys, `roofline` and `cpu_baseline` as numbers only, `also` as
    {leg: fraction-or-rate} scalars.  Everything else (notes, per-leg timing, how-it-was-measured prose) stays in the
    full record (FULL_RECORD, also on stderr).  BENCH_r05 failed to parse at 20 KB; this is held under 4 KB by test."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
           "vs_baseline", "dtype", "data")
    line = {k: full.get(k) for k in top}                                # the contract's own numbers: full precision
    cfg = full.get("config", {})
    line["config"] = {"workload": "BASELINE configs[1]: batched 32x32 forward DCT, %d int16 residual blocks per GPU resident in HBM"
                                  % cfg.get("blocks_per_gpu", 0),
                      "blocks_per_gpu": cfg.get("blocks_per_gpu"), "block_bytes_in_plus_out": cfg.get("block_bytes_in_plus_out"),
                      "sharding": cfg.get("sharding")}
    r = full.get("roofline") or {}
    line["roofline"] = {k: _sig(r.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_per_launch",
                                                     "kernel_ms_mean", "algorithmic_bytes_per_launch", "frac_of_same_box_copy", "frac_at_mean")}
    if r.get("frac_by_rank"):
        line["roofline"]["frac_by_rank"] = [_sig(v, 4) for v in r["frac_by_rank"]]
    c = full.get("cpu_baseline")
    line["cpu_baseline"] = None if not c else {k: _sig(c.get(k)) for k in ("value", "unit", "cores", "kind", "single_thread_blocks_per_s",
                                                                             "host_cpu", "gpu_output_bit_exact_vs_cpu")}
    if c:
        line["cpu_baseline"]["sample"] = "all %d blocks of the GPU batch, same inputs, %s pinned threads" % (cfg.get("blocks_per_gpu", 0), c.get("cores"))
    if "secondary" in full:
        line["secondary"] = {k: _sig(v) for k, v in full["secondary"].items()}
    also, failed = {}, []
    for name, leg in (full.get("also") or {}).items():
        _leg_scalars(name, leg, also, failed)
    if full.get("also") is not None:
        line["also"] = also
        line["checks_failed"] = failed
        if full["also"].get("node_layer_error"):
            line["node_layer_error"] = str(full["also"]["node_layer_error"])[:200]
        if full["also"].get("rccl_by_rank"):
            line["rccl_ranks"] = len(full["also"]["rccl_by_rank"])
    line["output_checksum_sum_i16"] = full.get("output_checksum_sum_i16")
    line["device"] = full.get("device")
    line["full_record"] = FULL_RECORD
    if "error" in full:
        line["error"] = full["error"]
    return line


def emit(result, json_fd):
    """rank 0: the full record to FULL_RECORD (cwd; also gpurun_out/ when it exists), the compact line -- alone -- to stdout.
    stderr only gets a pointer: a consumer that reads both streams must not find a 20 KB line after the compact one."""
    full = json.dumps(result)
    for path in (FULL_RECORD, os.path.join("gpurun_out", FULL_RECORD) if os.path.isdir("gpurun_out") else None):
        if path:
            try:
                with open(path, "w") as f:
                    f.write(full + "\n")
            except OSError as e:
                sys.stderr.write("bench.py: could not write %s: %s\n" % (path, e))
    sys.stderr.write("bench.py: full record (%d bytes) in %s\n" % (len(full), os.path.abspath(FULL_RECORD)))
    sys.stderr.flush()
    os.write(json_fd, (json.dumps(compact_record(result), separators=(",", ":")) + "\n").encode())


def spawn_ranks(args):
    """plain `python bench.py --gpus N`: become the launcher -- the command line the driver uses, one rank per GPU; torchrun picks and
    HOLDS the rendezvous port itself (--standalone: no bind-close-reuse race); rank 0's JSON line goes to this
    process's stdout, the exit status is the job's"""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")))


def headline(b, x, z, same_box):
    head = b.timed_leg(lambda: b.codec.dct32_fwd_dev(x.ptr, z.ptr, b.n_dct, b.stream))
    pmc, pmc_src = traffic_of_headline(b)
    result = {
        "metric": "dct32_fwd_blocks_per_s", "value": b.rate(head, b.n_dct), "unit": "blocks/s", "n_gpus": b.world,
        "steps": b.K, "warmup": b.W, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: batched 32x32 forward DCT, %d random 9-bit residual blocks per GPU "
                               "resident in HBM (inverse, fused fwd+inv, 8x8 SATD and the other configs under 'also')" % b.n_dct,
                   "blocks_per_gpu": b.n_dct, "block_bytes_in_plus_out": DCT_BYTES_PER_BLOCK,
                   "arithmetic": "int16 data as two int8 planes x int8 coefficients on v_mfma_i32_32x32x32_i8, int32 accumulate",
                   "sharding": "contiguous shard per rank, no data-path collective",
                   "clock_prewarm_launches": head["clock_prewarm_launches"],
                   "timing": "every leg: own clock pre-warm, W warm-up launches, K launches barrier-to-barrier; kernel_ms = 10 % trimmed mean of the HIP-event "
                             "durations recorded on the launching stream around those same K launches (mean and median next to it)"},
        "roofline": b.roofline(head, DCT_BYTES_PER_BLOCK, b.n_dct, pmc.get("dct32_fwd_bytes_per_launch"), pmc_src),
        "hip_runtime": {"version": b.hip.version(), "path": b.hip.paths[0], "copies_in_process": len(loaded_libraries("libamdhip64.so")),
                        "torch_imported": "torch" in sys.modules},
    }
    result["roofline"]["same_box"] = same_box
    return result, pmc, pmc_src


def main():
    args = parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.traffic_child:
        traffic_child(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    # stdout carries exactly ONE line, the JSON: libraries that chat on stdout (RCCL prints a version banner
    # when a communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    b = Bench(args)
    rank, world, codec = b.rank, b.world, b.codec

    # ---- inputs resident in HBM: this rank's slice of the one seeded stream -----------------
    x, z = b.dev(b.n_dct * 2048), b.dev(b.n_dct * 2048)
    codec.fill_residual_dev(x.ptr, b.n_dct * 1024, DCT_SEED, rank * b.n_dct * 1024, b.stream)
    b.hip.device_sync()
    same_box = leg_same_box(b, x, z)
    result, pmc, pmc_src = headline(b, x, z, same_box)
    # ---- checksum of the forward output across ranks (validates the sharded run) ----------------
    z_host = z.download(np.int16, b.n_dct * 1024)
    result["output_checksum_sum_i16"] = b.sum_over_ranks_i64(int(z_host.astype(np.int64).sum()))

    if not args.no_also:
        also, me = {}, {}
        also.update(leg_dct32_inverse_and_fused(b, x, z, pmc, pmc_src))
        also["satd8x8"] = leg_satd(b, pmc, pmc_src)
        if not args.no_me:
            also.update(leg_motion_search(b, me))
        if not args.no_transform_set:
            also["transform_set"] = leg_transform_set(b, x)
            also["fused_from_tiles"] = leg_fused_from_tiles(b)
            also["front_end_and_sad"] = leg_front_end_and_sad(b)
            also["intra32"] = leg_intra(b)
            if not args.no_autotune:
                also["autotune"] = leg_autotuned(b, x, z)
        if rank == 0 and world == 1 and not args.no_host_api:
            also["host_api"] = leg_host_api(b, min(b.n_dct, 1 << 17))
        result["also"] = also
        result["secondary"] = {"metric": "satd8x8_blocks_per_s", "value": also["satd8x8"]["value"], "unit": "blocks/s",
                               "roofline_frac": also["satd8x8"]["roofline"]["frac"],
                               "frac_of_same_box_read": also["satd8x8"]["roofline"]["frac_of_same_box_read"]}
        if (not b.share or os.environ.get("X266HIP_RCCL_LIB")) and args.stream8k > 0:
            run_node_legs_under_watchdog(b, result, also, x, z, me, json_fd)
        also["rccl_libraries_in_process"] = loaded_libraries("librccl")

    # ---- CPU baseline for the headline leg (rank 0, N = 1 only) ------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        x_host = x.download(np.int16, b.n_dct * 1024).reshape(b.n_dct, 1024)
        base, exact = cpu_baseline_dct(x_host, z_host.reshape(b.n_dct, 1024))
        result["cpu_baseline"] = base
        if not exact:
            result["error"] = "GPU output differs from the CPU reference"
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        result["device"] = b.info["name"].strip()
        sys.stdout.flush()
        emit(result, json_fd)
    if b.dist is not None:
        if result.get("also", {}).get("node_layer_error"):               # peers may be stuck in a collective this rank left: no orderly shutdown
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        b.dist.destroy_process_group()
        # the library brought ROCm's HIP runtime, torch (control plane) was imported after it and bundles a second copy under the same SONAME: leave without
        # running two sets of exit handlers over one runtime's state (a CPU session in that import order ended in "double free or corruption" at exit)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
