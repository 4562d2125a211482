"""GPU: the north-star floor as a regression guard -- BASELINE.json asks for >= 1e8 DCT32 blocks/s at >= 70 % of the
relevant roofline on one MI355X.  Measured values are ~1.6e9 blocks/s and 0.83 (profiles/r01_bench.json); the
thresholds here leave room for the slowest box seen (-8 %) and only catch real regressions."""
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
        ms = min(cd.time_kernel(op, din.ptr, dout.ptr, units, 20) for _ in range(3))
        got[name] = (units / ms * 1e3, units * unit_bytes / (ms * 1e-3) / HBM_PEAK)
    assert got["fwd"][0] >= 1e8 and got["fwd"][1] >= 0.70, got     # the north star itself
    assert got["inv"][1] >= 0.65 and got["satd"][1] >= 0.62, got
