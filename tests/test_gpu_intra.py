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
