"""GPU parity: xIntra32PredictDev (32x32 intra prediction, HEVC 35 modes; parity unpinned upstream)
against the oracle -- every mode on ramps, noise, flat and extreme borders, the WIP testbench's own
stimulus, shared and per-block reference sets, ragged counts."""
import numpy as np
import pytest

import x266_amd
from _util import intra_refs_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    return x266_amd.Codec(0)


def test_every_mode_every_border(codec, oracle):
    refs = intra_refs_np(40, 0x2468)
    modes = np.tile(np.arange(35, dtype=np.uint8), refs.shape[0])
    idx = np.repeat(np.arange(refs.shape[0], dtype=np.uint32), 35)
    got = codec.intra32_predict(refs, modes, idx)
    want = oracle.intra32_predict(refs, modes, idx)
    bad = np.argwhere((got != want).any(axis=1)).ravel()
    assert bad.size == 0, [(int(idx[b]), int(modes[b])) for b in bad[:10]]


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 63, 257, 1001])
def test_ragged_counts_one_set_per_block(codec, oracle, n):
    refs = intra_refs_np(max(n, 1), 77 + n)[:n]
    modes = ((np.arange(n) * 7 + 3) % 35).astype(np.uint8)
    got = codec.intra32_predict(refs, modes) if n else np.zeros((0, 1024), np.uint8)
    assert got.shape == (n, 1024)
    if n:
        assert np.array_equal(got, oracle.intra32_predict(refs, modes))


def test_rejects_bad_arguments(codec):
    d = codec.alloc(4096)
    with pytest.raises(x266_amd.X266Error):
        codec.intra32_predict_dev(d.ptr + 1, d.ptr, 0, d.ptr + 1024, 1)     # misaligned reference sets
    with pytest.raises(x266_amd.X266Error):
        codec.intra32_predict_dev(0, d.ptr, 0, d.ptr + 1024, 1)


def test_large_batch_all_modes(codec, oracle):
    """64 k predictions (64 MiB of output): all 35 modes of 1872 borders; compared on a strided sample
    and by checksum of the whole output against the oracle."""
    nref = 1872
    refs = intra_refs_np(nref, 4242)
    modes = np.tile(np.arange(35, dtype=np.uint8), nref)
    idx = np.repeat(np.arange(nref, dtype=np.uint32), 35)
    got = codec.intra32_predict(refs, modes, idx)
    want = oracle.intra32_predict(refs, modes, idx)
    assert int(got.astype(np.uint64).sum()) == int(want.astype(np.uint64).sum())
    assert np.array_equal(got, want)


def _sources(oracle, refs, seed):
    """Source blocks that make the decision non-trivial: each is one mode's prediction plus noise."""
    n = refs.shape[0]
    modes = ((np.arange(n) * 11 + seed) % 35).astype(np.uint8)
    clean = oracle.intra32_predict(refs, modes).astype(np.int64)
    from _util import splitmix64
    noise = ((splitmix64(seed, 0, n * 1024) >> np.uint64(21)) & np.uint64(15)).astype(np.int64).reshape(n, 1024) - 7
    return np.clip(clean + noise, 0, 255).astype(np.uint8), modes


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 37, 300])
def test_mode_decision_costs_and_winner(codec, oracle, n):
    refs = intra_refs_np(max(n, 5), 1000 + n)[:n]
    src, planted = _sources(oracle, refs, 3 + n)
    costs, best = codec.intra32_costs(refs, src)
    ocosts, obest = oracle.intra32_costs(refs, src)
    assert np.array_equal(costs, ocosts)                     # every mode of every block
    assert np.array_equal(best, obest)                       # same winner, same tie-break
    smooth = np.arange(n) >= 5                                # sets 0..4 are the flat / extreme / noise borders
    if smooth.any():
        assert np.mean(best[smooth] == planted[smooth]) > 0.5


def test_mode_decision_extremes_and_ties(codec, oracle):
    refs = intra_refs_np(5, 31)
    src = np.zeros((5, 1024), np.uint8)
    src[1] = 255                                              # flat border 255 + flat source 255: all costs 0, mode 0 wins
    src[3] = np.where((np.arange(1024) // 32 + np.arange(1024) % 32) % 2 == 0, 255, 0)
    src[4] = np.arange(1024) % 251
    costs, best = codec.intra32_costs(refs, src)
    ocosts, obest = oracle.intra32_costs(refs, src)
    assert np.array_equal(costs, ocosts) and np.array_equal(best, obest)
    assert np.all(costs[1] == 0) and best[1] == 0
    assert costs.max() < (1 << 26)                            # the (cost << 6 | mode) key cannot overflow


def test_mode_decision_is_prediction_plus_satd(codec, oracle):
    """Composition check against the two GPU kernels it fuses: predictions, residual sub-blocks, xSatd8x8BatchDev."""
    refs = intra_refs_np(9, 77)
    src, _ = _sources(oracle, refs, 5)
    costs, _ = codec.intra32_costs(refs, src)
    modes = np.tile(np.arange(35, dtype=np.uint8), 9)
    idx = np.repeat(np.arange(9, dtype=np.uint32), 35)
    pred = codec.intra32_predict(refs, modes, idx).reshape(9, 35, 32, 32).astype(np.int16)
    diff = src.reshape(9, 1, 32, 32).astype(np.int16) - pred
    sub = diff.reshape(9, 35, 4, 8, 4, 8).transpose(0, 1, 2, 4, 3, 5).reshape(-1, 64)
    sat = codec.satd8x8(sub).reshape(9, 35, 16).sum(axis=2)
    assert np.array_equal(costs, sat.astype(np.uint32))


# ---- prediction -> residual -> forward DCT32 in one kernel (xIntra32ResidualDct32Dev, round 5) -----------------------------------
def _want_coefficients(oracle, refs, modes, src, idx=None):
    """the composition it fuses, on the host: oracle predictor -> src - pred (9-bit) -> the pinned forward transform"""
    pred = oracle.intra32_predict(refs, modes, idx)
    return oracle.dct32_fwd(src.astype(np.int16) - pred.astype(np.int16), threads=8)


def test_residual_dct32_every_mode_every_border(codec, oracle):
    refs = intra_refs_np(40, 0x1357)
    modes = np.tile(np.arange(35, dtype=np.uint8), refs.shape[0])
    idx = np.repeat(np.arange(refs.shape[0], dtype=np.uint32), 35)
    src = np.random.RandomState(5).randint(0, 256, (modes.shape[0], 1024)).astype(np.uint8)
    src[::7] = 255                                                     # extremes of the 9-bit residual
    src[3::7] = 0
    got = codec.intra32_residual_dct32(refs, modes, src, idx)
    want = _want_coefficients(oracle, refs, modes, src, idx)
    bad = np.argwhere((got != want).any(axis=1)).ravel()
    assert bad.size == 0, [(int(idx[b]), int(modes[b])) for b in bad[:10]]


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 6, 7, 9, 63, 258, 1003])
def test_residual_dct32_ragged_counts(codec, oracle, n):
    """a wave takes four blocks: every remainder, one set per block (no index table)"""
    refs = intra_refs_np(max(n, 1), 99 + n)[:n]
    modes = ((np.arange(n) * 11 + 2) % 35).astype(np.uint8)
    src = np.random.RandomState(n).randint(0, 256, (n, 1024)).astype(np.uint8)
    got = codec.intra32_residual_dct32(refs, modes, src) if n else np.zeros((0, 1024), np.int16)
    assert got.shape == (n, 1024)
    if n:
        assert np.array_equal(got, _want_coefficients(oracle, refs, modes, src))


def test_residual_dct32_equals_the_three_kernels_it_fuses(codec, oracle):
    """xIntra32PredictDev -> xResidual (host subtraction here) -> xDct32FwdBatchDev on 20 k blocks, device to device"""
    nref, n = 600, 21000
    refs = intra_refs_np(nref, 777)
    rs = np.random.RandomState(9)
    modes = rs.randint(0, 35, n).astype(np.uint8)
    idx = rs.randint(0, nref, n).astype(np.uint32)
    src = rs.randint(0, 256, (n, 1024)).astype(np.uint8)
    got = codec.intra32_residual_dct32(refs, modes, src, idx)
    pred = codec.intra32_predict(refs, modes, idx)
    assert np.array_equal(got, codec.dct32_fwd(src.astype(np.int16) - pred.astype(np.int16)))
    assert np.array_equal(got[:2000], _want_coefficients(oracle, refs, modes[:2000], src[:2000], idx[:2000]))


def test_residual_dct32_rejects_bad_arguments(codec):
    d = codec.alloc(8192)
    with pytest.raises(x266_amd.X266Error):
        codec.intra32_residual_dct32_dev(d.ptr, d.ptr, 0, d.ptr + 1024 + 4, d.ptr + 4096, 1)   # misaligned source
    with pytest.raises(x266_amd.X266Error):
        codec.intra32_residual_dct32_dev(d.ptr, d.ptr, 0, d.ptr + 1024, 0, 1)                  # NULL coefficients


def test_intra_kernels_beyond_4_gib(codec, oracle):
    """Maximum sizes: 2^22 + 5 predictions (4 GiB + of predicted pixels) and 2^21 + 5 fused blocks (4 GiB + of coefficients): samples at the start,
    across the 2^32-byte boundary of the output and at the ragged end are bit-exact with the oracle."""
    n_sets = 4096
    n_pred, n_fused = (1 << 22) + 5, (1 << 21) + 5
    d_refs = codec.alloc(n_sets * 144)
    codec.fill_residual_dev(d_refs.ptr, n_sets * 72, 0x1A7)                      # any bytes are a valid reference set
    refs = d_refs.download(np.uint8, n_sets * 144).reshape(n_sets, 144)[:, :129]
    q = np.arange(n_pred, dtype=np.uint64)
    modes = ((q * 13 + q // 35) % 35).astype(np.uint8)
    idx = ((q * 2654435761) % n_sets).astype(np.uint32)
    d_modes, d_idx = codec.alloc(n_pred), codec.alloc(n_pred * 4)
    d_modes.upload(modes)
    d_idx.upload(idx)
    d_out = codec.alloc(max(n_pred * 1024, n_fused * 2048))                      # predictions, then (same bytes) the fused kernel's coefficients
    d_src = codec.alloc(n_fused * 1024)
    codec.fill_residual_dev(d_src.ptr, n_fused * 512, 0x51C)

    def sample(buf, first, count, dtype, unit):
        out = np.empty(count * unit, dtype)
        codec._check(codec.L.xHipMemcpyD2H(codec.ctx, out.ctypes.data, buf.ptr + first * unit * out.itemsize, out.nbytes), "D2H")
        return out.reshape(count, unit)

    codec.intra32_predict_dev(d_refs.ptr, d_modes.ptr, d_idx.ptr, d_out.ptr, n_pred)
    codec.stream_sync()
    for first, count in [(0, 40), ((1 << 22) - 20, 25), (n_pred - 36, 36)]:     # prediction 2^22 starts at byte 2^32
        want = oracle.intra32_predict(refs, modes[first:first + count], idx[first:first + count])
        assert np.array_equal(sample(d_out, first, count, np.uint8, 1024), want), first
    codec.intra32_residual_dct32_dev(d_refs.ptr, d_modes.ptr, d_idx.ptr, d_src.ptr, d_out.ptr, n_fused)
    codec.stream_sync()
    for first, count in [(0, 40), ((1 << 21) - 20, 25), (n_fused - 36, 36)]:    # block 2^21's coefficients start at byte 2^32
        src = sample(d_src, first, count, np.uint8, 1024)
        want = _want_coefficients(oracle, refs, modes[first:first + count], src, idx[first:first + count])
        assert np.array_equal(sample(d_out, first, count, np.int16, 1024), want), first
