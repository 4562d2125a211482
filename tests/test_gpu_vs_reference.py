"""GPU: the HIP path against the REAL reference, directly and in volume (missing #3).

oracle/_ref/libx266ref.so is src_tb/dct32.c + satd.c compiled where they lie (oracle/Makefile); it travels to the GPU box with the
snapshot.  Every other -m gpu test compares with the oracle (this repo's restatement, itself pinned to the reference in the build
container): valid, but one hop longer than needed.  Here the batch kernels' outputs are compared with the reference's own
partialButterfly32 x 2 (src_tb/dct32.c:197-198) on 1.2e5 mixed blocks, with its satd8x8 (src_tb/satd.c:31-118) on 1.3e6 blocks, and
the 1-D pass with partialButterfly32 itself at the two shifts dct32_genNew uses (src_tb/dct32.c:180-181).  Skipped (with the reason)
where the prebuilt reference is absent."""
import numpy as np
import pytest

from _util import dct_edge_blocks, extremes_np, fullrange_np, residual_np, satd_edge_blocks

pytestmark = pytest.mark.gpu


def _mixed(n_tri, n_full, n_ext, unit, seed, edge):
    parts = [residual_np(n_tri * unit, seed).reshape(n_tri, unit), fullrange_np(n_full * unit, seed + 0x1000).reshape(n_full, unit),
             extremes_np(n_ext * unit, seed + 0x2000).reshape(n_ext, unit), edge]
    x = np.concatenate(parts).astype(np.int16)
    return x[np.random.RandomState(seed).permutation(x.shape[0])]          # the kinds interleaved, not in runs


def test_dct32_forward_batch_equals_the_reference_on_120k_mixed_blocks(codec, reference):
    x = _mixed(80000, 30000, 10000, 1024, 0x51, dct_edge_blocks()[0])
    assert x.shape[0] >= 100000
    got = codec.dct32_fwd(x)                                              # host-pointer call: staging pipeline + the batch kernel
    want = reference.dct32_fwd(x)
    assert np.array_equal(got, want), "first differing block %d" % int(np.argmax((got != want).any(axis=1)))


def test_satd8x8_batch_equals_the_reference_on_1m3_mixed_blocks(codec, reference):
    x = _mixed(900000, 300000, 100000, 64, 0x52, satd_edge_blocks()[0])
    assert x.shape[0] >= 1000000
    want = reference.satd8x8(x)
    for variant in (0, 1, 3):                                             # by batch size / staged kernel / LDS-DMA kernel
        codec.set_option("satd_variant", variant)
        got = codec.satd8x8(x)
        assert np.array_equal(got, want), (variant, int(np.argmax(got != want)))
    codec.set_option("satd_variant", 0)


@pytest.mark.parametrize("shift", [4, 11])
def test_one_dimensional_pass_equals_partialButterfly32(codec, reference, shift):
    """xDct32PassDev = partialButterfly32(src, dst, shift, 32) with its transposed store, 1000 mixed blocks per shift"""
    x = _mixed(600, 300, 84, 1024, 0x53 + shift, dct_edge_blocks()[0])[:1000]
    got = codec.dct32_pass(x, shift)
    for b in range(x.shape[0]):
        assert np.array_equal(got[b], reference.dct32_pass(x[b], shift)), (shift, b)


def test_fused_forward_half_equals_the_reference(codec, reference):
    """the coefficients xDct32FwdInvBatchDev writes are the reference's (its inverse half has no upstream counterpart)"""
    x = _mixed(6000, 1500, 500, 1024, 0x54, dct_edge_blocks()[0])
    n = x.shape[0]
    d_in, d_coef, d_rec = codec.alloc(x.nbytes), codec.alloc(x.nbytes), codec.alloc(x.nbytes)
    d_in.upload(x)
    codec.dct32_fwd_inv_dev(d_in.ptr, d_coef.ptr, d_rec.ptr, n)
    codec.stream_sync()
    assert np.array_equal(d_coef.download(np.int16, n * 1024).reshape(n, 1024), reference.dct32_fwd(x))
