"""GPU (-m gpu): full-search SATD motion estimation (BASELINE configs[2]) through
xSatd8x8SearchDev against the oracle's brute-force search (per-candidate cost =
the reference's satd8x8 on the difference block).  Bit-exact costs, identical
winners including the tie-break."""
import numpy as np
import pytest

from _util import me_frames, splitmix64

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,rng,tile_rows", [
    (64, 16, 4, 2), (64, 48, 8, 1), (64, 48, 8, 2), (64, 48, 8, 4), (136, 72, 16, 2), (136, 72, 16, 4),
    (72, 40, 1, 2), (200, 24, 33, 2), (8, 8, 5, 2), (128, 64, 64, 2), (24, 136, 64, 4), (136, 72, 16, 8), (72, 136, 64, 8), (64, 64, 3, 8),
    (64, 48, 8, 8), (200, 24, 33, 1), (24, 136, 64, 1), (264, 200, 12, 4), (72, 72, 2, 4), (80, 80, 7, 2), (96, 40, 31, 8)])
def test_every_candidate_cost_and_winner(codec, oracle, w, h, rng, tile_rows):
    pad = rng + 3
    cur, refp = me_frames(w, h, pad, 100 + w + h + rng, mv=(min(rng, 3), -min(rng, 2)))
    saved = {k: codec.get_option(k) for k in ("me_tile_rows",)}
    codec.set_option("me_tile_rows", tile_rows)
    try:
        mv, cost, costs = codec.satd_search(cur, refp, pad, rng, want_costs=True)
        mv2, cost2, _ = codec.satd_search(cur, refp, pad, rng)          # the search-only kernel (no cost map)
    finally:
        for k, v in saved.items():
            codec.set_option(k, v)
    omv, ocost, ocosts = oracle.satd_search(cur, refp, pad, rng, threads=8, want_costs=True)
    assert np.array_equal(costs, ocosts)                 # all (2R+1)^2 costs of every block
    assert np.array_equal(cost, ocost)
    assert np.array_equal(mv, omv)                       # same winner => same tie-break
    assert np.array_equal(cost2, ocost) and np.array_equal(mv2, omv)


def test_tie_break_is_first_in_raster_order(codec, oracle):
    """Flat frames: every candidate costs the same, the first one (-R,-R) must win."""
    w, h, rng, pad = 64, 32, 6, 8
    cur = np.full((h, w), 77, np.uint8)
    refp = np.full((h + 2 * pad, w + 2 * pad), 80, np.uint8)
    mv, cost, _ = codec.satd_search(cur, refp, pad, rng)
    assert np.all(mv == [-rng, -rng]) and np.all(cost == (3 * 64 + 2) >> 2)
    omv, ocost, _ = oracle.satd_search(cur, refp, pad, rng)
    assert np.array_equal(mv, omv) and np.array_equal(cost, ocost)


def test_extreme_pixels(codec, oracle):
    """0/255 checkerboards drive the coefficients to their largest magnitudes."""
    w, h, rng, pad = 64, 32, 5, 8
    yy, xx = np.mgrid[0:h, 0:w]
    cur = np.where((xx + yy) % 2 == 0, 255, 0).astype(np.uint8)
    r = splitmix64(9, 0, (h + 2 * pad) * (w + 2 * pad))
    refp = np.where((r >> np.uint64(13)) & np.uint64(1), 255, 0).astype(np.uint8).reshape(h + 2 * pad, w + 2 * pad)
    mv, cost, costs = codec.satd_search(cur, refp, pad, rng, want_costs=True)
    omv, ocost, ocosts = oracle.satd_search(cur, refp, pad, rng, threads=4, want_costs=True)
    assert np.array_equal(costs, ocosts) and np.array_equal(mv, omv) and np.array_equal(cost, ocost)
    assert costs.max() <= 32640


def test_planted_motion_is_found(codec):
    w, h, rng, pad = 256, 128, 24, 24
    cur, refp = me_frames(w, h, pad, 4242, mv=(-11, 7), noise=3)
    mv, cost, _ = codec.satd_search(cur, refp, pad, rng)
    assert (mv == [-11, 7]).all(axis=1).mean() > 0.95


def test_argument_errors(codec):
    L = codec.L
    buf = codec.alloc(1 << 16)
    assert L.xSatd8x8SearchDev(codec.ctx, buf.ptr, 64, buf.ptr, 200, 60, 32, 8, buf.ptr, None, None) < 0    # width % 8
    assert L.xSatd8x8SearchDev(codec.ctx, buf.ptr, 64, buf.ptr, 200, 64, 32, 65, buf.ptr, None, None) < 0   # range
    assert L.xSatd8x8SearchDev(codec.ctx, buf.ptr, 64, buf.ptr, 64, 64, 32, 8, buf.ptr, None, None) < 0     # ref stride
    assert L.xSatd8x8SearchDev(codec.ctx, None, 64, buf.ptr, 200, 64, 32, 8, buf.ptr, None, None) < 0


def test_full_frame_4k_every_block(codec, oracle):
    """BASELINE configs[2] at full size, checked WHOLE: 3840x2160, window +-64 -- all 129 600 (mv, cost) records against the
    oracle's brute force (2.16e9 SATDs; the oracle is built -O3 and runs on every host thread: tens of seconds).  Rounds 1-5
    checked 32 of the 270 block rows."""
    w, h, rng, pad = 3840, 2160, 64, 64
    cur, refp = me_frames(w, h, pad, 2160, mv=(5, -3), noise=4)
    mv, cost, _ = codec.satd_search(cur, refp, pad, rng)
    assert mv.shape == (129600, 2) and (mv == [5, -3]).all(axis=1).mean() > 0.9
    omv, ocost, _ = oracle.satd_search(cur, refp, pad, rng, threads=min(270, oracle.hw_threads()))
    assert np.array_equal(cost, ocost)
    assert np.array_equal(mv, omv)                                      # same winner everywhere => same tie-break everywhere


def test_full_frame_4k_cost_maps(codec, oracle):
    """The cost-map form of the same 4K search ("64 blocks, every candidate's cost"): the launch writes all
    129 600 x 16 641 costs (8.6 GB, stays in HBM); 64 blocks spread over the frame -- corners, edges, tile boundaries,
    interior -- are downloaded and compared candidate by candidate with the oracle, and for EVERY block of the frame the
    map's first minimum must be the (mv, cost) record the search-only kernel returns."""
    w, h, rng, pad = 3840, 2160, 64, 64
    span, ncand, bxn = 2 * rng + 1, (2 * rng + 1) ** 2, w // 8
    nb = bxn * (h // 8)
    cur, refp = me_frames(w, h, pad, 2160, mv=(5, -3), noise=4)
    dc, dr, db, dcost = codec.alloc(cur.nbytes), codec.alloc(refp.nbytes), codec.alloc(nb * 8), codec.alloc(nb * ncand * 4)
    dc.upload(cur)
    dr.upload(refp)
    codec.satd_search_dev(dc.ptr, w, dr.ptr + pad * refp.shape[1] + pad, refp.shape[1], w, h, rng, db.ptr, dcost.ptr)
    codec.stream_sync()
    raw = db.download(np.uint8, nb * 8)
    mv = raw.view(np.int16).reshape(nb, 4)[:, :2]
    cost = raw.view(np.uint32).reshape(nb, 2)[:, 1]
    mv0, cost0, _ = codec.satd_search(cur, refp, pad, rng)              # the search-only kernel
    assert np.array_equal(mv, mv0) and np.array_equal(cost, cost0)

    def maps(first_block, count):                                       # [count, ncand] of the device's cost map
        out = np.empty((count, ncand), np.uint32)
        codec._check(codec.L.xHipMemcpyD2H(codec.ctx, out.ctypes.data, dcost.ptr + first_block * ncand * 4, out.nbytes), "xHipMemcpyD2H")
        return out
    # (block row, first block column) of eight 8-block runs = 64 blocks
    runs = [(0, 0), (0, bxn - 8), (269, 0), (269, bxn - 8), (7, 100), (8, 236), (135, 0), (201, 333)]
    for by, bx in runs:
        got = maps(by * bxn + bx, 8)
        _, _, want = oracle.satd_search(cur[by * 8:by * 8 + 8, bx * 8:bx * 8 + 64],
                                        refp[by * 8:by * 8 + 8 + 2 * pad, bx * 8:bx * 8 + 64 + 2 * pad], pad, rng, want_costs=True)
        assert np.array_equal(got, want), (by, bx)
    step = 2048                                                         # 136 MB of map per download
    for b0 in range(0, nb, step):
        m = maps(b0, min(step, nb - b0))
        k = m.argmin(axis=1)                                            # numpy: first occurrence = raster order = the tie-break
        assert np.array_equal(m[np.arange(len(k)), k], cost[b0:b0 + len(k)]), b0
        assert np.array_equal(np.stack([k % span - rng, k // span - rng], axis=1), mv[b0:b0 + len(k)]), b0


@pytest.mark.parametrize("w,h,rng", [(640, 360, 64), (1280, 136, 64), (3840, 136, 64), (3840, 544, 64), (200, 72, 20), (1920, 1080, 64)])
def test_automatic_tile_height_changes_nothing_but_the_time(codec, w, h, rng):
    """"me_tile_rows" = 0 (the default) picks 8-, 4- or 2-row tiles from the frame size; whatever it picks, mv, cost and
    the SAD search's winners equal those of explicitly chosen tile heights (each checked against the oracle above)."""
    assert codec.get_option("me_tile_rows") == 0                     # the default is the automatic choice
    h -= h % 8
    cur, refp = me_frames(w, h, rng, 31 + w, mv=(3, -2), noise=3)
    got = codec.satd_search(cur, refp, rng, rng)
    got_sad = codec.satd_search(cur, refp, rng, rng, metric="sad")
    try:
        for tr in (8, 4, 2):
            codec.set_option("me_tile_rows", tr)
            mv, cost, _ = codec.satd_search(cur, refp, rng, rng)
            assert np.array_equal(mv, got[0]) and np.array_equal(cost, got[1]), tr
            mv, cost, _ = codec.satd_search(cur, refp, rng, rng, metric="sad")
            assert np.array_equal(mv, got_sad[0]) and np.array_equal(cost, got_sad[1]), tr
    finally:
        codec.set_option("me_tile_rows", 0)
