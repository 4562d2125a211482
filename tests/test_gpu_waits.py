"""GPU (-m gpu): the hand-counted `s_waitcnt vmcnt(N)` of the LDS-DMA kernels, checked by execution.

x266_amd/libx266hip_waits0.so is the same source built with -DX266_WAIT_ALL (make -C x266_amd/csrc waits0; __graft_entry__.build() does
it): every counted wait there waits for EVERYTHING.  A count that is too high in the product would read an LDS slot before its DMA has
landed; the two libraries must therefore write the same bytes -- on runs far longer than the pipelines' depth, ragged tails, single
blocks, with and without the coefficient output, and under every launch shape the autotuner may pick."""
import os

import numpy as np
import pytest

from _util import ROOT

pytestmark = pytest.mark.gpu
W0 = os.path.join(ROOT, "x266_amd", "libx266hip_waits0.so")


@pytest.fixture(scope="module")
def waits0():
    import x266_amd
    assert os.path.exists(W0), "x266_amd/libx266hip_waits0.so is not built (make -C x266_amd/csrc waits0)"
    c = x266_amd.Codec(0, library=W0)
    yield c
    c.close()


def _fused(cd, n, with_coef, seed, opts):
    for k, v in opts.items():
        cd.set_option(k, v)
    x, z, r = cd.alloc(n * 2048), cd.alloc(n * 2048), cd.alloc(n * 2048)
    cd.fill_residual_dev(x.ptr, n * 1024, seed)
    cd.dct32_fwd_inv_dev(x.ptr, z.ptr if with_coef else 0, r.ptr, n)
    cd.stream_sync()
    out = (z.download(np.int16, n * 1024) if with_coef else None, r.download(np.int16, n * 1024))
    for k in opts:
        cd.set_option(k, 0)
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 4101, 70001])
@pytest.mark.parametrize("with_coef", [True, False])
@pytest.mark.parametrize("bpw", [0, 1, 3, 16, 40])
def test_fused_forward_inverse(codec, waits0, n, with_coef, bpw):
    """blocks per wave from below the DMA depth (2) to far above it: prologue, steady state and drain of the counted pipeline"""
    opts = {"dct32_fwdinv_blocks_per_wave": bpw, "adaptive_per_wave": 0}
    a, b = _fused(codec, n, with_coef, 0x61 + n, opts), _fused(waits0, n, with_coef, 0x61 + n, opts)
    codec.set_option("adaptive_per_wave", 1); waits0.set_option("adaptive_per_wave", 1)
    assert np.array_equal(a[1], b[1])
    if with_coef:
        assert np.array_equal(a[0], b[0])


def test_fused_depth_three_shapes_of_the_autotuner(waits0):
    """the DEPTH = 3 instantiation only runs when "autotune" picks it: force every candidate by timing on a box-sized batch and compare
    tuned output (whatever was kept) against the all-waiting library's default"""
    import x266_amd
    n = (1 << 18) + 3
    tuned = x266_amd.Codec(0)
    try:
        tuned.set_option("autotune", 1)                                  # the tuning call itself launches all eight candidates over the outputs
        a = _fused(tuned, n, True, 0x71, {})
        assert len(tuned.autotune_report()["dct32_fwd_inv"]["ms"]) == 8
    finally:
        tuned.close()
    b = _fused(waits0, n, True, 0x71, {})
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, (1 << 17) + 7])
@pytest.mark.parametrize("gpw", [0, 1, 2, 5, 17])
def test_satd_lds_dma_batch(codec, waits0, n, gpw):
    outs = []
    for cd in (codec, waits0):
        cd.set_option("satd_variant", 3); cd.set_option("satd_groups_per_wave", gpw); cd.set_option("adaptive_per_wave", 0)
        d, s = cd.alloc(n * 128), cd.alloc(n * 4)
        cd.fill_residual_dev(d.ptr, n * 64, 0x81 + n)
        cd.satd8x8_dev(d.ptr, s.ptr, n)
        cd.stream_sync()
        outs.append(s.download(np.uint32, n))
        cd.set_option("satd_variant", 0); cd.set_option("satd_groups_per_wave", 0); cd.set_option("adaptive_per_wave", 1)
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("w,h", [(16, 16), (144, 16), (208, 112), (1920, 1088)])
def test_satd_from_tiles_lds_dma(codec, waits0, w, h):
    nt = (w // 16) * (h // 16)
    outs = []
    for cd in (codec, waits0):
        cd.set_option("satd_variant", 3)
        a, b, s = cd.alloc(nt * 512), cd.alloc(nt * 512), cd.alloc(nt * 16)
        cd.fill_residual_dev(a.ptr, nt * 256, 0x91)
        cd.fill_residual_dev(b.ptr, nt * 256, 0x92)
        cd.satd8x8_from_tiles_dev(a.ptr, b.ptr, w, h, s.ptr)
        cd.stream_sync()
        outs.append(s.download(np.uint32, nt * 4))
        cd.set_option("satd_variant", 0)
    assert np.array_equal(outs[0], outs[1])
