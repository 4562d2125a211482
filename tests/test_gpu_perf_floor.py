"""GPU: performance floors as regression guards, in BOX-NORMALISED units (round 4).

Boxes of the pool differ by 3-10 % in what a plain stream reaches -- more than any effect worth guarding -- so every HBM-bound
kernel is measured next to this box's own arithmetic-free streams (xHipMemCeilingDev: copy / read / write in the launch shape
that measured fastest for each) and the floors are fractions of THOSE, a few per cent under what rounds 3-4 measured on several
boxes (profiles/r04_perf_floor_calibration.txt).  The north star's own absolute line (>= 1e8 DCT32 blocks/s at >= 70 % of the
8 TB/s roofline) is asserted as well.  The two motion searches are VALU-bound: their floor is the v_sad issue time."""
import statistics
import time

import numpy as np
import pytest

import x266_amd
from x266_amd.node import Node
from _dev import Dev

pytestmark = pytest.mark.gpu

HBM_PEAK = 8.0e12


def _median_ms(cd, fn, warm=60, reps=15):
    for _ in range(warm):
        fn()
    cd.stream_sync()
    ev = [cd.event_create() for _ in range(reps + 1)]
    for i in range(reps):
        cd.event_record(ev[i])
        fn()
    cd.event_record(ev[reps])
    ms = statistics.median(cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(reps))
    for e in ev:
        cd.event_destroy(e)
    return ms


STREAM_KIND = {"copy": (0, 2), "read": (1, 1), "write": (2, 1)}     # xHipMemCeilingDev kind, buffer passes moved


def _best_fraction(cd, fn, kernel_bytes, stream, x, z, nbytes, attempts=3, warm=60, reps=15):
    """kernel rate / this box's stream rate, the stream measured right BEFORE and right AFTER the kernel (mean of the two), best of
    `attempts`: the chip's clocks move with what ran in the seconds before (a whole test suite, here), so a normaliser taken once
    at module start made 1 run in 3 of the full suite fail on the same box.  A regression fails every attempt; a transient does not."""
    kind, passes = STREAM_KIND[stream]

    def stream_rate():
        ms = _median_ms(cd, lambda: cd.mem_ceiling_dev(kind, x.data_ptr(), z.data_ptr(), nbytes), warm=40, reps=9)
        return passes * nbytes / (ms * 1e-3)
    best, best_ms = 0.0, None
    for _ in range(attempts):
        s0 = stream_rate()
        ms = _median_ms(cd, fn, warm=warm, reps=reps)
        s1 = stream_rate()
        frac = kernel_bytes / (ms * 1e-3) / (0.5 * (s0 + s1))
        if frac > best:
            best, best_ms = frac, ms
    return best, best_ms


@pytest.fixture(scope="module")
def bench():
    """2 GiB of residual, its outputs, and this box's streams over the same buffers (bytes per second)"""
    cd = x266_amd.Codec(0)
    D = Dev(cd)
    n = 1 << 20
    x = D.empty(n * 1024, np.int16)
    z = D.empty_like(x)
    r = D.empty_like(x)
    cd.fill_residual_dev(x.data_ptr(), x.numel(), 0x266)
    cd.stream_sync()
    nbytes = n * 2048
    box = {}
    for kind, name, moved in ((0, "copy", 2 * nbytes), (1, "read", nbytes), (3, "read_no_store", nbytes), (2, "write", nbytes)):
        ms = _median_ms(cd, lambda: cd.mem_ceiling_dev(kind, x.data_ptr(), z.data_ptr(), nbytes), warm=120)   # clocks need ~50 ms of load
        box[name] = moved / (ms * 1e-3)
    print("\nthis box: " + ", ".join("%s %.2f TB/s" % (k, v / 1e12) for k, v in box.items()))
    return cd, n, x, z, r, box


def test_streams_of_this_box_are_sane(bench):
    """the normalisers themselves: a box whose copy stream is under 6 TB/s (0.75 of the spec) is broken, not slow"""
    box = bench[5]
    assert box["copy"] >= 6.0e12 and box["read"] >= 6.0e12 and box["write"] >= 5.0e12, box
    assert box["read_no_store"] >= 0.98 * box["read"], box


def test_headline_kernels_against_this_box(bench):
    cd, n, x, z, r, box = bench
    nbytes = n * 2048
    ns = 1 << 24                                            # x holds 2^24 SATD blocks worth of samples
    out = Dev(cd).empty(ns, np.int32)
    legs = {"fwd": (lambda: cd.dct32_fwd_dev(x.data_ptr(), z.data_ptr(), n), n, 4096, "copy"),
            "inv": (lambda: cd.dct32_inv_dev(z.data_ptr(), r.data_ptr(), n), n, 4096, "copy"),
            "satd": (lambda: cd.satd8x8_dev(x.data_ptr(), out.data_ptr(), ns), ns, 132, "read"),
            "fused": (lambda: cd.dct32_fwd_inv_dev(x.data_ptr(), z.data_ptr(), r.data_ptr(), n), n, 6144, "copy")}
    rel, got = {}, {}
    for name, (fn, units, unit_bytes, stream) in legs.items():
        rel[name], ms = _best_fraction(cd, fn, units * unit_bytes, stream, x, z, nbytes)
        got[name] = (units / ms * 1e3, units * unit_bytes / (ms * 1e-3))
    # Two more allocation SETS (keep round 4's floors and average the placement out instead of lowering them): where a process's buffers
    # land moves a kernel by 2-9 % against the stream on the same buffers (profiles/r05_placement.txt), and attempts on the SAME buffers do not
    # average that out -- fresh buffers are another draw, the best draw is what the kernel can do, a regression is below the floor on every one.
    D = Dev(cd)
    for _ in range(2):
        x2 = D.empty(n * 1024, np.int16)
        z2, r2 = D.empty_like(x2), D.empty_like(x2)
        out2 = D.empty(ns, np.int32)
        cd.fill_residual_dev(x2.data_ptr(), x2.numel(), 0x266)
        legs2 = {"fwd": (lambda: cd.dct32_fwd_dev(x2.data_ptr(), z2.data_ptr(), n), n, 4096, "copy"),
                 "inv": (lambda: cd.dct32_inv_dev(z2.data_ptr(), r2.data_ptr(), n), n, 4096, "copy"),
                 "satd": (lambda: cd.satd8x8_dev(x2.data_ptr(), out2.data_ptr(), ns), ns, 132, "read"),
                 "fused": (lambda: cd.dct32_fwd_inv_dev(x2.data_ptr(), z2.data_ptr(), r2.data_ptr(), n), n, 6144, "copy")}
        for name, (fn, units, unit_bytes, stream) in legs2.items():
            if rel[name] >= FLOORS[name] + 0.02:
                continue                                               # comfortably above already
            frac, ms = _best_fraction(cd, fn, units * unit_bytes, stream, x2, z2, nbytes, attempts=2)
            if frac > rel[name]:
                rel[name] = frac
                got[name] = max(got[name], (units / ms * 1e3, units * unit_bytes / (ms * 1e-3)))
        del x2, z2, r2, out2
    print("\nfractions of this box's streams: " + ", ".join("%s %.3f" % kv for kv in rel.items()) +
          " | of 8 TB/s: " + ", ".join("%s %.3f" % (k, v[1] / HBM_PEAK) for k, v in got.items()))
    assert got["fwd"][0] >= 1e8 and got["fwd"][1] / HBM_PEAK >= 0.70, got       # the north star itself
    assert got["inv"][1] / HBM_PEAK >= 0.70 and got["satd"][1] / HBM_PEAK >= 0.70, got
    assert rel["fwd"] >= FLOORS["fwd"] and rel["inv"] >= FLOORS["inv"], rel
    assert rel["fused"] >= FLOORS["fused"] and rel["satd"] >= FLOORS["satd"], rel


# fractions of the box's own copy / read / write stream; measured values and boxes in profiles/r04_perf_floor_calibration.txt.  Round 5: where a process's buffers
# land moves a kernel by 2-9 % against the stream on the same buffers (profiles/r05_placement.txt) and three attempts on the SAME buffers do not average that out:
# the headline kernels therefore take the best of up to three allocation SETS against round 4's floors (forward 0.97-0.99, inverse 0.93-0.96, fused 0.86-0.90, SATD batch 0.87-0.94 seen)
FLOORS = {"fwd": 0.96, "inv": 0.92, "fused": 0.82, "satd": 0.88, "tiles_fwd": 0.91, "tiles_inv": 0.90, "intra_write": 0.64}


def test_other_baseline_config_legs_against_this_box(bench):
    """configs[3] one-launch mixed CTU buffer, 32x32 intra prediction (write-bound), the one-rank 7680x4320 frame stream of configs[4]"""
    cd, n, x, z, r, box = bench
    got = {}
    D = Dev(cd)
    q = np.arange(n)
    cls = D.from_numpy(np.array([3, 2, 6, 1, 5, 0, 4], np.uint8)[(q + q // 4) % 7])
    nbytes = n * 2048
    for inv, name in ((0, "tiles_fwd"), (1, "tiles_inv")):
        got[name], _ = _best_fraction(cd, lambda: cd.transform_tiles_dev(inv, x.data_ptr(), z.data_ptr(), n, 0, cls.data_ptr()), n * 4096, "copy", x, z, nbytes)
    # intra prediction as bench.py runs it: every reference set predicted in all 35 modes (a mode decision's access pattern), 1 KiB written each
    n_refs = 59918
    n_pred = n_refs * 35
    refs = D.random_u8(n_refs * 144, 5)
    modes = D.from_numpy(np.tile(np.arange(35, dtype=np.uint8), n_refs))
    index = D.from_numpy(np.repeat(np.arange(n_refs, dtype=np.int32), 35))
    pred = D.empty(n_pred * 1024, np.uint8)
    got["intra_write"], _ = _best_fraction(cd, lambda: cd.intra32_predict_dev(refs.data_ptr(), modes.data_ptr(), index.data_ptr(), pred.data_ptr(), n_pred),
                                           n_pred * 1024, "write", x, z, nbytes, warm=20)
    del pred, refs
    # one-rank 7680x4320 frame stream through the node layer (needs an RCCL to open; any will do with one rank)
    node = Node.for_rank(0, 0, 1, Node.unique_id())
    nd, ns = (7680 // 32) * (4320 // 32), (7680 // 8) * (4320 // 8)
    st = node.frame_stream(7680, 4320)
    costs = [D.empty(ns, np.int32) for _ in range(5)]
    fin = [(x.data_ptr() + 2 * i * nd * 1024, x.data_ptr() + 2 * (8 + i) * ns * 64) for i in range(4)]     # slices of the 2 GiB input
    fout = [(z.data_ptr() + 2 * i * nd * 1024, costs[i].data_ptr()) for i in range(5)]

    def push(f):
        a, b = fin[f % 4]
        c, e = fout[f % 5]
        st.push([a, b], [c, e], producer_stream=st.next_slot_stream())
    for f in range(200):
        push(f)
    st.flush()
    us, copy_rate = 1e9, 0.0
    for _ in range(3):                                       # best of three runs, the copy stream measured between them
        ms = _median_ms(cd, lambda: cd.mem_ceiling_dev(0, x.data_ptr(), z.data_ptr(), nbytes), warm=40, reps=9)
        copy_rate = max(copy_rate, 2 * nbytes / (ms * 1e-3))
        for f in range(300):
            push(f)
        st.flush()
        t0 = time.perf_counter()
        for f in range(1000):
            push(f)
        st.flush()
        us = min(us, (time.perf_counter() - t0) / 1000 * 1e6)
    st.close()
    node.close()
    # the frame moves 201 MB (132.7 read + 68.5 written): against this box's copy stream
    got["stream8k_frac_of_copy_time"] = (nd * 4096 + ns * 132) / copy_rate / (us * 1e-6)
    print("\nfractions of this box's streams: " + ", ".join("%s %.3f" % kv for kv in got.items()) + " | stream8k %.1f us per frame" % us)
    assert got["tiles_fwd"] >= FLOORS["tiles_fwd"] and got["tiles_inv"] >= FLOORS["tiles_inv"], got
    assert got["intra_write"] >= FLOORS["intra_write"], got
    assert us <= 40.0 and got["stream8k_frac_of_copy_time"] >= 0.82, (us, got)


def test_motion_search_stays_above_its_floor_fraction():
    """configs[2]: one 3840x2160 frame, window +-64, against the v_sad_u16 issue floor (32 instructions per 64 candidates,
    4 cycles each, 1024 SIMDs at 2.4 GHz = 1.755 ms).  Rounds 3-4 measured 2.30-2.32 ms = 0.76 on the builder's boxes, 0.70-0.74
    on slower-clocking ones; the SAD search 1.21 ms = 0.72-0.73 of its v_sad_u8 floor (0.8777 ms)."""
    cd = x266_amd.Codec(0)
    w, h, rng = 3840, 2160, 64
    D = Dev(cd)
    cur = D.random_u8((h, w), 7)
    refp = D.random_u8((h + 2 * rng, w + 2 * rng), 8)
    best = D.empty((h // 8) * (w // 8) * 2, np.int32)
    org = refp.data_ptr() + rng * refp.stride(0) + rng
    ms = min(_median_ms(cd, lambda: cd.satd_search_dev(cur.data_ptr(), cur.stride(0), org, refp.stride(0), w, h, rng, best.data_ptr()), warm=30, reps=10) for _ in range(3))
    ms_sad = min(_median_ms(cd, lambda: cd.sad_search_dev(cur.data_ptr(), cur.stride(0), org, refp.stride(0), w, h, rng, best.data_ptr()), warm=30, reps=10) for _ in range(3))
    print("\nSATD search %.3f ms = %.3f of the v_sad_u16 floor; SAD search %.3f ms = %.3f of the v_sad_u8 floor" % (ms, 1.7554 / ms, ms_sad, 0.8777 / ms_sad))
    assert 1.7554 / ms >= 0.72, ms
    assert 0.8777 / ms_sad >= 0.68, ms_sad
