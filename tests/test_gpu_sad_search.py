"""GPU parity: xSad8x8SearchDev (full search with the SAD metric) against the oracle's brute force:
every candidate cost, the winner and the tie-break, over tile shapes, ranges and ragged frame sizes."""
import numpy as np
import pytest

import x266_amd
from _util import me_frames, splitmix64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    return x266_amd.Codec(0)


@pytest.mark.parametrize("w,h,rng,tile_rows", [
    (64, 16, 4, 2), (64, 48, 8, 1), (64, 48, 8, 2), (64, 48, 8, 4), (136, 72, 16, 2), (136, 72, 16, 4),
    (72, 40, 1, 2), (200, 24, 33, 2), (8, 8, 5, 2), (128, 64, 64, 2), (24, 136, 64, 4), (40, 40, 31, 4), (72, 72, 36, 4), (64, 64, 3, 2)])
def test_every_candidate_cost_and_winner(codec, oracle, w, h, rng, tile_rows):
    pad = rng + 3
    cur, refp = me_frames(w, h, pad, 300 + w + h + rng, mv=(min(rng, 3), -min(rng, 2)))
    saved = codec.get_option("me_tile_rows")
    codec.set_option("me_tile_rows", tile_rows)
    try:
        mv, cost, costs = codec.satd_search(cur, refp, pad, rng, want_costs=True, metric="sad")
        mv2, cost2, _ = codec.satd_search(cur, refp, pad, rng, metric="sad")       # the search-only instantiation
    finally:
        codec.set_option("me_tile_rows", saved)
    omv, ocost, ocosts = oracle.satd_search(cur, refp, pad, rng, threads=8, want_costs=True, metric="sad")
    assert np.array_equal(costs, ocosts)
    assert np.array_equal(cost, ocost) and np.array_equal(mv, omv)
    assert np.array_equal(cost2, ocost) and np.array_equal(mv2, omv)


def test_ties_and_extremes(codec, oracle):
    w, h, rng, pad = 64, 32, 6, 8
    cur = np.full((h, w), 77, np.uint8)
    refp = np.full((h + 2 * pad, w + 2 * pad), 80, np.uint8)
    mv, cost, _ = codec.satd_search(cur, refp, pad, rng, metric="sad")
    assert np.all(mv == [-rng, -rng]) and np.all(cost == 3 * 64)           # flat frames: the first candidate wins
    yy, xx = np.mgrid[0:h, 0:w]
    cur = np.where((xx + yy) % 2 == 0, 255, 0).astype(np.uint8)
    r = splitmix64(19, 0, (h + 2 * pad) * (w + 2 * pad))
    refp = np.where((r >> np.uint64(13)) & np.uint64(1), 255, 0).astype(np.uint8).reshape(h + 2 * pad, w + 2 * pad)
    mv, cost, costs = codec.satd_search(cur, refp, pad, rng, want_costs=True, metric="sad")
    omv, ocost, ocosts = oracle.satd_search(cur, refp, pad, rng, threads=4, want_costs=True, metric="sad")
    assert np.array_equal(costs, ocosts) and np.array_equal(mv, omv) and np.array_equal(cost, ocost)
    assert costs.max() <= 64 * 255


def test_argument_errors(codec):
    d = codec.alloc(1 << 16)
    with pytest.raises(x266_amd.X266Error):
        codec.sad_search_dev(d.ptr + 1, 64, d.ptr + 4096, 80, 64, 16, 4, d.ptr + 32768)   # misaligned current frame
    with pytest.raises(x266_amd.X266Error):
        codec.sad_search_dev(d.ptr, 64, d.ptr + 4096, 80, 64, 16, 65, d.ptr + 32768)      # range > 64


def test_full_frame_4k_every_block(codec, oracle):
    """The SAD search at BASELINE configs[2]'s size (3840x2160, window +-64): all 129 600 records against the oracle's
    brute force (rounds 3-5: 20 of the 270 block rows)."""
    w, h, rng, pad = 3840, 2160, 64, 64
    cur, refp = me_frames(w, h, pad, 2161, mv=(-7, 4), noise=4)
    mv, cost, _ = codec.satd_search(cur, refp, pad, rng, metric="sad")
    assert (mv == [-7, 4]).all(axis=1).mean() > 0.9
    omv, ocost, _ = oracle.satd_search(cur, refp, pad, rng, threads=min(270, oracle.hw_threads()), metric="sad")
    assert np.array_equal(cost, ocost) and np.array_equal(mv, omv)
