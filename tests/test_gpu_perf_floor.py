"""GPU: the north-star floor as a regression guard -- BASELINE.json asks for >= 1e8 DCT32 blocks/s at >= 70 % of the
relevant roofline on one MI355X.  Measured: forward ~1.6e9 blocks/s = 0.83, inverse 0.79, SATD batch 0.77 (0.71 on the
slowest box seen); the north-star fraction is asserted for all three, after a clock pre-warm of each kernel."""
import pytest

import x266_amd
from x266_amd._lib import OP_DCT32_FWD, OP_DCT32_INV, OP_SATD8X8

pytestmark = pytest.mark.gpu

HBM_PEAK = 8.0e12


def test_headline_kernels_stay_above_the_target_fraction():
    cd = x266_amd.Codec(0)
    n = 1 << 20
    din, dout = cd.alloc(n * 2048), cd.alloc(n * 2048)
    cd.fill_residual_dev(din.ptr, n * 1024, 0x266)
    cd.stream_sync()
    cd.time_kernel(OP_DCT32_FWD, din.ptr, dout.ptr, n, 120)          # clocks need ~50 ms of load
    got = {}
    for name, op, units, unit_bytes in (("fwd", OP_DCT32_FWD, n, 4096), ("inv", OP_DCT32_INV, n, 4096), ("satd", OP_SATD8X8, 1 << 24, 132)):
        cd.time_kernel(op, din.ptr, dout.ptr, units, 120)            # every kernel gets its own pre-warm
        ms = min(cd.time_kernel(op, din.ptr, dout.ptr, units, 20) for _ in range(3))
        got[name] = (units / ms * 1e3, units * unit_bytes / (ms * 1e-3) / HBM_PEAK)
    assert got["fwd"][0] >= 1e8 and got["fwd"][1] >= 0.70, got     # the north star itself
    assert got["inv"][1] >= 0.70 and got["satd"][1] >= 0.70, got


def test_motion_search_stays_above_its_floor_fraction():
    """configs[2]: one 3840x2160 frame, window +-64, against the v_sad_u16 issue floor (32 instructions per 64 candidates,
    4 cycles each, 1024 SIMDs at 2.4 GHz = 1.755 ms).  Round 2 measured 2.5 ms = 0.70; the guard is the VERDICT's 0.65
    less 5 % for a slow box."""
    import statistics
    import torch
    cd = x266_amd.Codec(0)
    w, h, rng = 3840, 2160, 64
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    cur = torch.randint(0, 256, (h, w), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
    refp = torch.randint(0, 256, (h + 2 * rng, w + 2 * rng), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
    best = torch.empty((h // 8) * (w // 8) * 2, dtype=torch.int32, device="cuda")
    org = refp.data_ptr() + rng * refp.stride(0) + rng
    fn = lambda: cd.satd_search_dev(cur.data_ptr(), cur.stride(0), org, refp.stride(0), w, h, rng, best.data_ptr())
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    ev = [cd.event_create() for _ in range(11)]
    for i in range(10):
        cd.event_record(ev[i])
        fn()
    cd.event_record(ev[10])
    ms = statistics.median(cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(10))
    assert 1.7554 / ms >= 0.62, ms
