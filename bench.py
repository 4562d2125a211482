#!/usr/bin/env python3
"""bench.py -- throughput of the x266 DCT32 / SATD hot path on MI355X.

Contract (one JSON line on stdout from rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[1]: 1,048,576 synthetic 9-bit residual blocks of
32x32 int16 per GPU, resident in HBM before the timed region (values a-b, a,b
uniform bytes -- the reference's stimulus distribution, src_tb/dct32.c:191-193 --
from SplitMix64 seed 0x266).  A "step" is one forward 2-D DCT32 pass over the
batch (xDct32FwdBatchDev through the C ABI).  `value` is whole-job forward
blocks/s over all ranks; the inverse transform and the 8x8 SATD residual batch
(2^24 blocks, seed 0x267) are measured the same way and reported under "also".

Independent blocks shard across ranks with no data-path collective (weak
scaling: every rank owns its own 1 Mi-block slice of the one seeded stream);
torch.distributed (RCCL) carries only the barrier, the max-over-ranks time and a
checksum reduction.

"roofline": algorithmic bytes per launch (4096 B per DCT block, 132 B per SATD
block) / mean kernel duration measured with HIP events on
the launching stream, against the 8 TB/s HBM3E peak.
"cpu_baseline": the reference C path on this node's host cores (oracle/_ref =
the real src_tb/dct32.c when its prebuilt .so is present, else the oracle's
restatement), bounded sample, rank 0 at N = 1 only.  The oracle is used here and
nowhere in the product.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_BYTES_PER_S = 8.0e12          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
DCT_BLOCKS_PER_GPU = 1 << 20           # BASELINE configs[1]
SATD_BLOCKS_PER_GPU = 1 << 24          # 2 GiB of 8x8 residual blocks
DCT_BYTES_PER_BLOCK = 4096             # 2048 read + 2048 written   
SATD_BYTES_PER_BLOCK = 132             # 128 read + 4 written
DCT_SEED, SATD_SEED = 0x266, 0x267


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed launches first; the chip needs ~50 ms of load to reach steady clocks (profiles/r01_clock_warmup.txt)")
    ap.add_argument("--dct-blocks", type=int, default=DCT_BLOCKS_PER_GPU)
    ap.add_argument("--satd-blocks", type=int, default=SATD_BLOCKS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the inverse / SATD legs")
    ap.add_argument("--stream8k", type=int, default=0, metavar="FRAMES",
                    help="also time BASELINE configs[4]: 7680x4320 frames, DCT32+SATD batches scattered from rank 0 and gathered back")
    ap.add_argument("--no-me", action="store_true", help="skip the motion-search leg")
    ap.add_argument("--no-transform-set", action="store_true", help="skip the transform-set leg")
    return ap.parse_args()


def cpu_baseline_dct(x_host, gpu_out_host):
    """Reference C path timed on the host cores (rank 0, N = 1).  Returns the
    cpu_baseline object; also checks the GPU output against it bit-for-bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _util import Oracle, Reference, ref_path

    orc = Oracle()
    cores = orc.hw_threads()
    n = x_host.shape[0]
    # single thread, small sample
    n1 = min(n, 16384)
    t = time.perf_counter()
    orc.dct32_fwd(x_host[:n1], threads=1)
    single = n1 / (time.perf_counter() - t)

    have_ref = os.path.exists(ref_path())
    out = np.empty_like(x_host)
    if have_ref:
        ref = Reference()
        bounds = np.linspace(0, n, cores + 1).astype(np.int64)

        def work(i):
            b, e = int(bounds[i]), int(bounds[i + 1])
            if e > b:
                ref.lib.ref_dct32_fwd(ctypes.c_void_p(x_host[b:e].ctypes.data), ctypes.c_void_p(out[b:e].ctypes.data),
                                      ctypes.c_ulong(e - b))

        ths = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
        t = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t
        kind = "reference"
    else:
        t = time.perf_counter()
        out = orc.dct32_fwd(x_host, threads=cores)
        dt = time.perf_counter() - t
        kind = "port"
    exact = bool(np.array_equal(out, gpu_out_host))
    # secondary figure: the same restatement built -O3 -march=native ON THIS HOST
    native = None
    try:
        import glob
        import subprocess
        import tempfile
        so = os.path.join(tempfile.gettempdir(), "liborc_native_%d.so" % os.getpid())
        srcs = sorted(glob.glob(os.path.join(ROOT, "oracle", "*_oracle.c")))
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=gnu11", "-fPIC", "-shared", "-o", so] + srcs + ["-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        nat = ctypes.CDLL(so)
        tmp = np.empty_like(x_host)
        t = time.perf_counter()
        nat.orc_dct32_fwd_mt(ctypes.c_void_p(x_host.ctypes.data), ctypes.c_void_p(tmp.ctypes.data), ctypes.c_size_t(n), cores)
        dt_n = time.perf_counter() - t
        if np.array_equal(tmp, out):
            native = n / dt_n
        os.unlink(so)
    except Exception:
        native = None
    return {
        "value": n / dt, "unit": "blocks/s", "cores": cores, "kind": kind,
        "sample": "%d of the %d blocks of the GPU batch (same inputs), one contiguous shard per thread, -O2" % (n, n),
        "single_thread_blocks_per_s": single,
        "port_O3_march_native_all_cores_blocks_per_s": native,
        "gpu_output_bit_exact_vs_cpu": exact,
    }, exact


def main():
    args = parse_args()
    import torch
    import x266_amd
    from x266_amd._lib import OP_DCT32_FWD, OP_DCT32_INV, OP_SATD8X8

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libx266hip has no CPU path")
    # Test hook (never set by the driver): X266_BENCH_SHARE_GPU=1 lets several ranks share the visible
    # GPUs with the control-plane collectives on gloo, so that the N > 1 code path -- shard offsets,
    # max-over-ranks timing, checksum reduction -- can be exercised on a one-GPU box.
    share = os.environ.get("X266_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    ctrl = "cuda"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
            ctrl = "cpu"
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    codec = x266_amd.Codec(local_rank)
    info_cu = codec.device_info()["cu_count"]
    stream = torch.cuda.current_stream().cuda_stream           # the stream every launch and event uses
    n_dct, n_satd = args.dct_blocks, args.satd_blocks

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=ctrl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- inputs resident in HBM: this rank's slice of the one seeded stream -----------------
    x = torch.empty(n_dct * 1024, dtype=torch.int16, device="cuda")
    z = torch.empty_like(x)
    codec.fill_residual_dev(x.data_ptr(), n_dct * 1024, DCT_SEED, rank * n_dct * 1024, stream)
    torch.cuda.synchronize()

    def run_leg(op, fn, d_in, d_out, n_units, steps, warmup):
        for _ in range(warmup):
            fn(d_in, d_out, n_units, stream)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(d_in, d_out, n_units, stream)
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        # per-launch kernel duration: HIP events recorded on the launching stream itself
        kernel_ms = codec.time_kernel(op, d_in, d_out, n_units, steps, stream)
        return wall, kernel_ms

    # Clock pre-warm, separate from the W warm-up steps of the contract: the chip needs ~50 ms of load to reach
    # steady clocks (profiles/r01_clock_warmup.txt).  With a small --warmup the timed steps would otherwise sit
    # on the ramp; with the default W = 100 this adds nothing.
    prewarm = max(0, 100 - args.warmup)
    for _ in range(prewarm):
        codec.dct32_fwd_dev(x.data_ptr(), z.data_ptr(), n_dct, stream)
    wall, k_ms = run_leg(OP_DCT32_FWD, codec.dct32_fwd_dev, x.data_ptr(), z.data_ptr(), n_dct, args.steps, args.warmup)
    value = world * n_dct * args.steps / wall

    def roofline(bytes_per_unit, n_units, kernel_ms, traffic=None):
        achieved = bytes_per_unit * n_units / (kernel_ms * 1e-3)
        return {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_BYTES_PER_S, "traffic": traffic,
                "kernel_ms_per_launch": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_unit * n_units}

    # HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (collected
    # with this same command on this workload; FETCH_SIZE x2 gfx950 correction applied there).
    # They only describe the default workload size.
    pmc = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and n_dct == DCT_BLOCKS_PER_GPU and n_satd == SATD_BLOCKS_PER_GPU:
        try:
            pmc = json.load(open(tpath))
        except Exception:
            pmc = {}
    traffic = pmc.get("dct32_fwd_bytes_per_launch")

    result = {
        "metric": "dct32_fwd_blocks_per_s", "value": value, "unit": "blocks/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: batched 32x32 forward DCT, %d random 9-bit residual blocks per GPU "
                               "resident in HBM (inverse + 8x8 SATD legs under 'also')" % n_dct,
                   "blocks_per_gpu": n_dct, "block_bytes_in_plus_out": DCT_BYTES_PER_BLOCK,
                   "arithmetic": "int16 data as two int8 planes x int8 coefficients on v_mfma_i32_32x32x32_i8, int32 accumulate",
                   "sharding": "contiguous shard per rank, no data-path collective",
                   "clock_prewarm_launches": prewarm},
        "roofline": roofline(DCT_BYTES_PER_BLOCK, n_dct, k_ms, traffic),
    }

    # ---- checksum of the forward output across ranks (validates the sharded run) ----------------
    csum = int(z.view(torch.int16).to(torch.int64).sum().item())
    if dist is not None:
        t = torch.tensor([csum], dtype=torch.int64, device=ctrl)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        csum = int(t.item())
    result["output_checksum_sum_i16"] = csum

    # ---- also: inverse DCT32 and SATD residual batch ---------------------------------------------
    if not args.no_also:
        also = {}
        r = torch.empty_like(x)
        wall_i, k_ms_i = run_leg(OP_DCT32_INV, codec.dct32_inv_dev, z.data_ptr(), r.data_ptr(), n_dct, args.steps, args.warmup)
        also["dct32_inv"] = {"value": world * n_dct * args.steps / wall_i, "unit": "blocks/s",
                             "ms_per_step": wall_i / args.steps * 1e3,
                             "roofline": roofline(DCT_BYTES_PER_BLOCK, n_dct, k_ms_i, pmc.get("dct32_inv_bytes_per_launch")),
                             "parity": "unpinned (no inverse in the reference); bit-exact vs this repo's oracle"}
        err = (r[: 4096 * 1024].to(torch.int32) - x[: 4096 * 1024].to(torch.int32)).abs().max().item()
        also["dct32_inv"]["roundtrip_max_abs_err"] = int(err)
        # fused forward + inverse: coefficients and reconstruction from one pass (6144 B per block)
        z2 = torch.empty_like(x)

        def fused(d_in, d_out, n_units, st):
            codec.dct32_fwd_inv_dev(d_in, z2.data_ptr(), d_out, n_units, st)
        for _ in range(args.warmup):
            fused(x.data_ptr(), r.data_ptr(), n_dct, stream)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fused(x.data_ptr(), r.data_ptr(), n_dct, stream)
        barrier()
        wall_f = max_over_ranks(time.perf_counter() - t0)
        also["dct32_fwd_inv_fused"] = {"value": world * n_dct * args.steps / wall_f, "unit": "blocks/s",
                                       "ms_per_step": wall_f / args.steps * 1e3,
                                       "hbm_frac": 6144.0 * n_dct * args.steps / wall_f / HBM_PEAK_BYTES_PER_S,
                                       "same_bytes_as_two_kernels": bool(torch.equal(z2, z)),
                                       "note": "wall-clock (launch gaps included); 2 KiB in, 2 + 2 KiB out per block"}
        del r, z2
        d = torch.empty(n_satd * 64, dtype=torch.int16, device="cuda")
        s = torch.empty(n_satd, dtype=torch.int32, device="cuda")
        codec.fill_residual_dev(d.data_ptr(), n_satd * 64, SATD_SEED, rank * n_satd * 64, stream)
        wall_s, k_ms_s = run_leg(OP_SATD8X8, codec.satd8x8_dev, d.data_ptr(), s.data_ptr(), n_satd, args.steps, args.warmup)
        also["satd8x8"] = {"value": world * n_satd * args.steps / wall_s, "unit": "blocks/s",
                           "ms_per_step": wall_s / args.steps * 1e3, "blocks_per_gpu": n_satd,
                           "roofline": roofline(SATD_BYTES_PER_BLOCK, n_satd, k_ms_s, pmc.get("satd8x8_bytes_per_launch"))}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from _util import Oracle
            orc = Oracle()
            ns = min(n_satd, 1 << 22)
            dh = d[: ns * 64].cpu().numpy()
            t0 = time.perf_counter()
            ref_s = orc.satd8x8(dh, threads=orc.hw_threads())
            dt = time.perf_counter() - t0
            also["satd8x8"]["cpu_baseline"] = {
                "value": ns / dt, "unit": "blocks/s", "cores": orc.hw_threads(), "kind": "port",
                "sample": "first %d blocks of the GPU batch" % ns,
                "gpu_output_bit_exact_vs_cpu": bool(np.array_equal(ref_s.astype(np.int32), s[:ns].cpu().numpy()))}
        del d, s

        # ---- BASELINE configs[2]: full-search SATD motion estimation, one 3840x2160 luma frame, window +-64
        if not args.no_me:
            w, h, rng = 3840, 2160, 64
            pad = rng
            g = torch.Generator(device="cuda")
            g.manual_seed(0x266 + rank)
            big = torch.randint(0, 256, (h + 2 * pad + 16, w + 2 * pad + 16), generator=g, device="cuda", dtype=torch.int32)
            # 5x5 box low-pass so that motion is findable (pooling, not conv: no MIOpen kernel search)
            sm = torch.nn.functional.avg_pool2d(big.float()[None, None], 5, stride=1, padding=2, count_include_pad=False)[0, 0]
            sm = ((sm - 128.0) * 3.0 + 128.0).clamp(0, 255).to(torch.uint8)
            cur = sm[pad + 8:pad + 8 + h, pad + 8:pad + 8 + w].contiguous()
            refp = sm[8 + 3:8 + 3 + h + 2 * pad, 8 - 5:8 - 5 + w + 2 * pad].contiguous()      # planted motion (5, -3)
            nb = (w // 8) * (h // 8)
            best = torch.empty(nb * 2, dtype=torch.int32, device="cuda")
            origin = refp.data_ptr() + pad * refp.stride(0) + pad

            def me(_a, _b, _n, st):
                codec.satd_search_dev(cur.data_ptr(), cur.stride(0), origin, refp.stride(0), w, h, rng, best.data_ptr(), 0, st)

            me_steps, me_warm = max(2, args.steps // 10), 2
            for _ in range(me_warm):
                me(0, 0, 0, stream)
            barrier()
            t0 = time.perf_counter()
            for _ in range(me_steps):
                me(0, 0, 0, stream)
            barrier()
            wall_m = max_over_ranks(time.perf_counter() - t0)
            ncand = nb * (2 * rng + 1) ** 2
            mv = best.view(torch.int16).view(nb, 4)[:, :2]
            found = float(((mv[:, 0] == 5) & (mv[:, 1] == -3)).float().mean().item())
            # VALU floor: 32 x v_sad_u16 (4 cycles per wave64 instruction, tools/alubench) per 64 candidates
            floor_s = ncand / 64 * 32 * 4 / (4 * info_cu * 2.4e9)
            also["satd8x8_me_search"] = {
                "value": world * ncand * me_steps / wall_m, "unit": "SATD/s", "ms_per_frame": wall_m / me_steps * 1e3,
                "frame": "%dx%d luma, 8x8 blocks, window +-%d (%d candidates per block)" % (w, h, rng, (2 * rng + 1) ** 2),
                "bound": "VALU issue (v_sad_u16), not HBM: ~18 MB of compulsory traffic per frame",
                "frac_of_v_sad_u16_floor": floor_s / (wall_m / me_steps),
                "planted_mv_found_fraction": found,
                "parity": "per-candidate cost pinned by satd8x8 (src_tb/satd.c); harness (order, tie-break, padding) unpinned"}
            # the same search with the SAD metric
            for _ in range(me_warm):
                codec.sad_search_dev(cur.data_ptr(), cur.stride(0), origin, refp.stride(0), w, h, rng, best.data_ptr(), 0, stream)
            barrier()
            t0 = time.perf_counter()
            for _ in range(me_steps):
                codec.sad_search_dev(cur.data_ptr(), cur.stride(0), origin, refp.stride(0), w, h, rng, best.data_ptr(), 0, stream)
            barrier()
            wall_sm = max_over_ranks(time.perf_counter() - t0)
            mv = best.view(torch.int16).view(nb, 4)[:, :2]
            also["sad8x8_me_search"] = {
                "value": world * ncand * me_steps / wall_sm, "unit": "SAD/s", "ms_per_frame": wall_sm / me_steps * 1e3,
                "frac_of_v_sad_u8_floor": (ncand / 64 * 16 * 4 / (4 * info_cu * 2.4e9)) / (wall_sm / me_steps),
                "planted_mv_found_fraction": float(((mv[:, 0] == 5) & (mv[:, 1] == -3)).float().mean().item()),
                "parity": "metric = sad() of riscv/programs/benchmarks/sad/sad.c at n = 8; harness unpinned, as for the SATD search"}
            del big, sm, cur, refp, best

        # ---- BASELINE configs[3]: the VVC transform set, 2 GiB of residual per class
        if not args.no_transform_set:
            ts = {}
            zt = torch.empty_like(x)                       # own output buffer: z still holds the headline leg's result
            for ttype, tname, inverse in ((0, "dct2", False), (1, "dst7", False), (0, "dct2_inv", True), (1, "dst7_inv", True)):
                for n in (4, 8, 16):
                    nblk = (n_dct * 1024) // (n * n)
                    if inverse:
                        fn = lambda a, b, cnt, st, tt=ttype, nn=n: codec.transform_inv_dev(tt, nn, a, b, cnt, 0, st)
                    else:
                        fn = lambda a, b, cnt, st, tt=ttype, nn=n: codec.transform_fwd_dev(tt, nn, a, b, cnt, 0, st)
                    for _ in range(3):
                        fn(x.data_ptr(), zt.data_ptr(), nblk, stream)
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        fn(x.data_ptr(), zt.data_ptr(), nblk, stream)
                    barrier()
                    wall_t = max_over_ranks(time.perf_counter() - t0)
                    ts["%s_%dx%d" % (tname, n, n)] = {
                        "value": world * nblk * args.steps / wall_t, "unit": "blocks/s",
                        "hbm_frac": 4.0 * n * n * nblk / (wall_t / args.steps) / HBM_PEAK_BYTES_PER_S}
            # per-CTU mixed batches: every 64x64 CTU's quadrants cycle through the four TU sizes; one call per
            # (type, size) class over an offset table into the shared residual / coefficient buffers
            n_ctu = (n_dct * 1024) // 4096
            q = torch.arange(n_ctu * 4, device="cuda", dtype=torch.int64)
            qbase, qkind = q * 1024, (q + q // 4) % 7          # quadrant -> one of the seven (type, size) classes
            mixed = []
            for kind, (tt, n) in enumerate(((0, 32), (0, 16), (1, 16), (0, 8), (1, 8), (0, 4), (1, 4))):
                base = qbase[qkind == kind]
                sub = torch.arange(1024 // (n * n), device="cuda", dtype=torch.int64) * (n * n)
                mixed.append((tt, n, (base[:, None] + sub[None, :]).reshape(-1).to(torch.int32).contiguous()))
            assert sum(o.numel() * nn * nn for _, nn, o in mixed) == n_ctu * 4096

            def ctu_pass():
                for tt, nn, o in mixed:
                    codec.transform_fwd_dev(tt, nn, x.data_ptr(), zt.data_ptr(), o.numel(), o.data_ptr(), stream)
            for _ in range(3):
                ctu_pass()
            barrier()
            t0 = time.perf_counter()
            for _ in range(max(2, args.steps // 4)):
                ctu_pass()
            barrier()
            wall_c = max_over_ranks(time.perf_counter() - t0) / max(2, args.steps // 4)
            per_ctu = {"value": world * n_ctu / wall_c, "unit": "CTUs/s", "hbm_frac": 4.0 * n_ctu * 4096 / wall_c / HBM_PEAK_BYTES_PER_S,
                       "layout": "64x64 CTUs whose 32x32 quadrants cycle through the seven classes (DCT-II 32/16/8/4, DST-VII 16/8/4), "
                                 "TUs of a quadrant contiguous; 7 calls per pass over offset tables", "ctus": n_ctu}
            # the same buffer in ONE launch: every quadrant is a tile with its own class (xTransformTilesDev)
            cls_of_kind = torch.tensor([3, 2, 6, 1, 5, 0, 4], device="cuda", dtype=torch.uint8)   # kinds above -> type*4 + log2N-2
            tile_cls = cls_of_kind[qkind].contiguous()
            for inv_flag, name in ((0, "per_ctu_one_launch"), (1, "per_ctu_one_launch_inverse")):
                for _ in range(3):
                    codec.transform_tiles_dev(inv_flag, x.data_ptr(), zt.data_ptr(), n_ctu * 4, 0, tile_cls.data_ptr(), stream)
                barrier()
                t0 = time.perf_counter()
                for _ in range(max(2, args.steps // 4)):
                    codec.transform_tiles_dev(inv_flag, x.data_ptr(), zt.data_ptr(), n_ctu * 4, 0, tile_cls.data_ptr(), stream)
                barrier()
                wall_o = max_over_ranks(time.perf_counter() - t0) / max(2, args.steps // 4)
                per_ctu[name] = {"value": world * n_ctu / wall_o, "unit": "CTUs/s", "hbm_frac": 4.0 * n_ctu * 4096 / wall_o / HBM_PEAK_BYTES_PER_S}
            del zt, mixed, q, qbase, qkind, tile_cls
            also["transform_set"] = {"classes": ts, "per_ctu_mixed": per_ctu, "parity": "unpinned upstream except DCT-II 32; bit-exact vs this repo's oracle",
                                     "note": "wall-clock rates (launch gaps included); 4*N*N algorithmic bytes per block"}
        # ---- fused front end: tiled cur/pred frames -> coefficients / costs, residual never in HBM
        if not args.no_transform_set:
            # 32768^2 luma: exactly 2^20 DCT32 blocks and 2^24 SATD blocks, i.e. the two-kernel legs launch the
            # headline kernels at the headline sizes (keeps rocprofv3's per-kernel averages comparable)
            fw, fh = 32768, 32768
            ntile = (fw // 16) * (fh // 16)
            gq = torch.Generator(device="cuda")
            gq.manual_seed(0x266)
            tcur = torch.randint(0, 256, (ntile * 512,), generator=gq, device="cuda", dtype=torch.uint8)
            tpred = torch.randint(0, 256, (ntile * 512,), generator=gq, device="cuda", dtype=torch.uint8)
            fcoef = torch.empty(fw * fh, dtype=torch.int16, device="cuda")
            fcost = torch.empty(fw * fh // 64, dtype=torch.int32, device="cuda")
            fres = torch.empty(fw * fh, dtype=torch.int16, device="cuda")
            fused = {}
            legs = (
                ("dct32_from_tiles", fw * fh // 1024, 4096,
                 lambda: codec.dct32_fwd_from_tiles_dev(tcur.data_ptr(), tpred.data_ptr(), fw, fh, fcoef.data_ptr(), stream)),
                ("dct32_residual_then_transform", fw * fh // 1024, None,
                 lambda: (codec.residual_luma_dev(tcur.data_ptr(), tpred.data_ptr(), fw, fh, 32, fres.data_ptr(), stream),
                          codec.dct32_fwd_dev(fres.data_ptr(), fcoef.data_ptr(), fw * fh // 1024, stream))),
                ("satd8x8_from_tiles", fw * fh // 64, 132,
                 lambda: codec.satd8x8_from_tiles_dev(tcur.data_ptr(), tpred.data_ptr(), fw, fh, fcost.data_ptr(), stream)),
                ("satd8x8_residual_then_cost", fw * fh // 64, None,
                 lambda: (codec.residual_luma_dev(tcur.data_ptr(), tpred.data_ptr(), fw, fh, 8, fres.data_ptr(), stream),
                          codec.satd8x8_dev(fres.data_ptr(), fcost.data_ptr(), fw * fh // 64, stream))))
            for name, units, bytes_per_unit, fn in legs:
                for _ in range(3):
                    fn()
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    fn()
                barrier()
                wall_f = max_over_ranks(time.perf_counter() - t0)
                fused[name] = {"value": world * units * args.steps / wall_f, "unit": "blocks/s"}
                if bytes_per_unit:
                    fused[name]["hbm_frac"] = bytes_per_unit * units / (wall_f / args.steps) / HBM_PEAK_BYTES_PER_S
            fused["note"] = ("%dx%d tiled frame pair (x266.cpp ref_block_t); fused kernels are bit-identical to the two-kernel paths "
                             "listed next to them (tests/test_gpu_tiles.py)" % (fw, fh))
            also["fused_from_tiles"] = fused
            del tcur, tpred, fcoef, fcost, fres

        # ---- frame container conversion, residual formation, SAD -- pure data movement, HBM-bound
        if not args.no_transform_set:
            fw2, fh2 = 16384, 16384                                      # 256 Mi luma samples
            npx = fw2 * fh2
            g = torch.Generator(device="cuda")
            g.manual_seed(0x77 + rank)
            ypl = torch.randint(0, 256, (npx,), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
            upl = ypl[: npx // 4].clone()
            vpl = ypl[npx // 4: npx // 2].clone()
            t_a = torch.zeros(npx * 2, dtype=torch.uint8, device="cuda")  # 512-byte tiles: 2 bytes per luma sample
            t_b = torch.zeros(npx * 2, dtype=torch.uint8, device="cuda")
            res2 = torch.empty(npx, dtype=torch.int16, device="cuda")
            sad_o = torch.empty(npx // 64, dtype=torch.int32, device="cuda")
            front = {}
            for name, nbytes, fn in (
                    ("conv_input_fmt", 3.0 * npx, lambda: codec.conv_input_fmt_dev(t_a.data_ptr(), ypl.data_ptr(), upl.data_ptr(), vpl.data_ptr(), fw2, fw2, fh2, stream)),
                    ("conv_output_420", 3.0 * npx, lambda: codec.conv_output_420_dev(t_a.data_ptr(), ypl.data_ptr(), fw2, upl.data_ptr(), vpl.data_ptr(), fw2 // 2, fw2, fh2, stream)),
                    ("residual_luma_32", 4.0 * npx, lambda: codec.residual_luma_dev(t_a.data_ptr(), t_b.data_ptr(), fw2, fh2, 32, res2.data_ptr(), stream)),
                    ("sad_8x8", 2.0 * npx + 4.0 * (npx // 64), lambda: codec.sad_dev(8, ypl.data_ptr(), t_b.data_ptr(), sad_o.data_ptr(), npx // 64, stream)),
                    ("sad_16x16", 2.0 * npx + 4.0 * (npx // 256), lambda: codec.sad_dev(16, ypl.data_ptr(), t_b.data_ptr(), sad_o.data_ptr(), npx // 256, stream)),
                    ("sad_64x64", 2.0 * npx + 4.0 * (npx // 4096), lambda: codec.sad_dev(64, ypl.data_ptr(), t_b.data_ptr(), sad_o.data_ptr(), npx // 4096, stream))):
                steps_f = max(2, args.steps // 4)
                for _ in range(5):
                    fn()
                barrier()
                t0 = time.perf_counter()
                for _ in range(steps_f):
                    fn()
                barrier()
                wall_ff = max_over_ranks(time.perf_counter() - t0) / steps_f
                front[name] = {"GBps": world * nbytes / wall_ff / 1e9, "hbm_frac": nbytes / wall_ff / HBM_PEAK_BYTES_PER_S,
                               "samples_per_s": world * npx / wall_ff}
            front["note"] = ("%dx%d frame; bytes = planes read + tile bytes written (conv), luma of both tile frames + int16 residual "
                             "(residual), both blocks + 4-byte result (sad)" % (fw2, fh2))
            also["front_end_and_sad"] = front
            del ypl, upl, vpl, t_a, t_b, res2, sad_o

        # ---- 32x32 intra prediction and mode decision (HEVC 35 modes; parity unpinned upstream)
        if not args.no_transform_set:
            g = torch.Generator(device="cuda")
            g.manual_seed(0x32 + rank)
            n_sets = 59918                                              # x 35 modes = 2 GiB of predictions
            refs_t = torch.randint(0, 256, (n_sets, 144), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
            modes_t = torch.arange(35, device="cuda", dtype=torch.uint8).repeat(n_sets)
            index_t = torch.arange(n_sets, device="cuda", dtype=torch.int32).repeat_interleave(35)
            pred_t = torch.empty(n_sets * 35 * 1024, dtype=torch.uint8, device="cuda")
            n_dec = 1 << 17
            src_t = torch.randint(0, 256, (n_dec * 1024,), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
            cost_t = torch.empty(n_dec * 35, dtype=torch.int32, device="cuda")
            bestm_t = torch.empty(n_dec, dtype=torch.uint8, device="cuda")
            intra = {}
            for name, units, fn in (
                    ("predict", n_sets * 35, lambda: codec.intra32_predict_dev(refs_t.data_ptr(), modes_t.data_ptr(), index_t.data_ptr(),
                                                                             pred_t.data_ptr(), n_sets * 35, stream)),
                    ("decide_35_modes", n_dec, lambda: codec.intra32_costs_dev(refs_t.data_ptr(), src_t.data_ptr(), cost_t.data_ptr(),
                                                                             bestm_t.data_ptr(), min(n_dec, n_sets), stream))):
                steps_i = max(2, args.steps // 4)
                for _ in range(5):
                    fn()
                barrier()
                t0 = time.perf_counter()
                for _ in range(steps_i):
                    fn()
                barrier()
                wall_i2 = max_over_ranks(time.perf_counter() - t0)
                units = units if name == "predict" else min(n_dec, n_sets)
                intra[name] = {"value": world * units * steps_i / wall_i2, "unit": "predictions/s" if name == "predict" else "blocks/s"}
            intra["predict"]["written_hbm_frac"] = 1024.0 * intra["predict"]["value"] / world / HBM_PEAK_BYTES_PER_S
            intra["decide_35_modes"]["satd8x8_per_s"] = intra["decide_35_modes"]["value"] * 35 * 16
            intra["parity"] = "unpinned upstream (src/mkIntra32-wip.bsv is a sketch without a model); bit-exact vs this repo's oracle"
            also["intra32"] = intra
            del refs_t, modes_t, index_t, pred_t, src_t, cost_t, bestm_t

        # ---- BASELINE configs[4] (on request): 8K frame stream, scatter -> kernels -> gather over RCCL
        if args.stream8k > 0 and ctrl == "cuda":
            from x266_amd.stream import FrameGeometry, PipelinedFrameStream
            geo = FrameGeometry(7680, 4320)
            dev = torch.device("cuda", local_rank)
            cur_stream = lambda: torch.cuda.current_stream().cuda_stream
            st8 = PipelinedFrameStream(
                geo, dev,
                lambda tin, tout, nblk: codec.dct32_fwd_dev(tin.data_ptr(), tout.data_ptr(), nblk, cur_stream()),
                lambda tin, tout, nblk: codec.satd8x8_dev(tin.data_ptr(), tout.data_ptr(), nblk, cur_stream()),
                dist=dist)
            fd = fs = None
            if rank == 0:
                fd = torch.empty(geo.dct_blocks * 1024, dtype=torch.int16, device=dev)
                fs = torch.empty(geo.satd_blocks * 64, dtype=torch.int16, device=dev)
                codec.fill_residual_dev(fd.data_ptr(), fd.numel(), DCT_SEED, 0, stream)
                codec.fill_residual_dev(fs.data_ptr(), fs.numel(), SATD_SEED, 0, stream)
            sums = []

            def sink8(f, coef, cost):
                if f == args.stream8k - 1:
                    sums.append(int(coef.to(torch.int64).sum().item()) + int(cost.to(torch.int64).sum().item()))
            st8.run(3, lambda f: (fd, fs), None)
            barrier()
            t0 = time.perf_counter()
            st8.run(args.stream8k, lambda f: (fd, fs), sink8)
            barrier()
            wall8 = max_over_ranks(time.perf_counter() - t0)
            also["stream8k"] = {
                "frames_per_s": args.stream8k / wall8, "ms_per_frame": wall8 / args.stream8k * 1e3,
                "frame": "7680x4320: %d DCT32 blocks (66.4 MB) + %d SATD blocks (66.4 MB)" % (geo.dct_blocks, geo.satd_blocks),
                "path": "rank 0 sends every peer its shard of frame f+1 and receives frame f-1's coefficients and costs "
                        "(batched isend/irecv on a side stream) while all ranks transform frame f"
                        if world > 1 else "single rank: device copies + kernels, no transfer",
                "link_bound": "one xGMI link ~153 GB/s => <= 7.5e7 DCT32 input blocks/s per peer (SURVEY.md 8e)"}
            if rank == 0:
                also["stream8k"]["output_checksum"] = sums[0] if sums else None
        result["also"] = also
        # BASELINE.json quotes two figures, 32x32 DCT blocks/s and 8x8 SATD blocks/s: the second one, lifted to the top level
        result["secondary"] = {"metric": "satd8x8_blocks_per_s", "value": also["satd8x8"]["value"], "unit": "blocks/s",
                               "roofline_frac": also["satd8x8"]["roofline"]["frac"]}

    # ---- CPU baseline for the headline leg (rank 0, N = 1 only) ------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, exact = cpu_baseline_dct(x.cpu().numpy().reshape(n_dct, 1024), z.cpu().numpy().reshape(n_dct, 1024))
        result["cpu_baseline"] = base
        if not exact:
            result["error"] = "GPU output differs from the CPU reference"
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        info = codec.device_info()
        result["device"] = info["name"].strip()
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
