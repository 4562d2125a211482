"""GPU: randomized differential test (hypothesis) -- random batch sizes, data mixes and launch options, every
kernel against the oracle.  Complements the hand-picked ragged counts of test_gpu_parity.py."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import x266_amd
from _util import extremes_np, fullrange_np, intra_refs_np, residual_np

pytestmark = pytest.mark.gpu

# X266_FUZZ_SCALE=k multiplies the example counts and turns random seeds on (campaigns run by hand: round 6, the final library -- all twelve tests incl. the tile stage with chroma and the whole-CTU launch, and the DCT / SATD batches over the 64 KiB host path -- 100x = 30 500 examples, clean (the tile-stage test alone also at 400x = 10 000 examples); round 5, the final library -- fused DCT32 and fused intra kernels, LDS-DMA from-tiles SATD, the re-shaped residual and inverse kernels -- 1000x = 255 000 examples, clean (and 300x before the last shape changes); round 4, after the SATD LDS-DMA kernel, the kernel prune, the new SAD and intra kernels, 100x = 25 500 examples, clean; round 3, after the tile kernel rewrite and with the three round-3 tests, 100x = 23 000 examples, clean; the round-2 campaign was
# 200x = 33 000 examples, clean); by default the examples are derived deterministically from the test body, so that a regular run of the
# suite does not depend on a random seed.
import os
_SCALE = int(os.environ.get("X266_FUZZ_SCALE", "1"))


def fuzz(n):
    return settings(max_examples=n * _SCALE, deadline=None, derandomize=_SCALE == 1, database=None,
                    suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow, HealthCheck.data_too_large])

OPTIONS = {
    "adaptive_per_wave": st.integers(0, 1),
    "dct32_variant": st.sampled_from([0, 2]),
    "satd_variant": st.sampled_from([0, 1, 2, 3]),
    "dct32_blocks_per_wave": st.integers(1, 9),
    "dct32_inv_blocks_per_wave": st.integers(1, 9),
    "dct32_fwdinv_blocks_per_wave": st.integers(1, 9),
    "satd_groups_per_wave": st.integers(0, 9),
    "dct32_wg_threads": st.sampled_from([0, 64, 128, 192, 256]),
    "satd_wg_threads": st.sampled_from([0, 64, 128, 192, 256]),
    "satd_lds_bytes_per_wave": st.sampled_from([0, 4096, 6144, 9216, 16384]),
}


@pytest.fixture(scope="module")
def codec():
    return x266_amd.Codec(0)


def _data(kind, n, unit, seed):
    if n == 0:
        return np.zeros((0, unit), np.int16)
    gen = (residual_np, fullrange_np, extremes_np)[kind]
    return gen(n * unit, seed).reshape(n, unit)


@fuzz(40)
@given(n=st.integers(0, 2500), kind=st.integers(0, 2), seed=st.integers(1, 1 << 30), opts=st.fixed_dictionaries(OPTIONS))
def test_dct_and_satd_random_sizes_and_options(codec, oracle, n, kind, seed, opts):
    saved = {k: codec.get_option(k) for k in opts}
    try:
        for k, v in opts.items():
            codec.set_option(k, v)
        x = _data(kind, n, 1024, seed)
        z = oracle.dct32_fwd(x, threads=8) if n else x
        assert np.array_equal(codec.dct32_fwd(x), z)
        assert np.array_equal(codec.dct32_inv(z), oracle.dct32_inv(z, threads=8) if n else z)
        d = _data(kind, n * 5 + (seed % 7), 64, seed + 1)
        assert np.array_equal(codec.satd8x8(d), oracle.satd8x8(d, threads=8) if d.shape[0] else np.zeros(0, np.uint32))
        if n:
            din, dco, dre = codec.alloc(n * 2048), codec.alloc(n * 2048), codec.alloc(n * 2048)
            din.upload(x)
            codec.dct32_fwd_inv_dev(din.ptr, dco.ptr, dre.ptr, n)
            codec.stream_sync()
            assert np.array_equal(dco.download(np.int16, n * 1024).reshape(n, 1024), z)
            assert np.array_equal(dre.download(np.int16, n * 1024).reshape(n, 1024), oracle.dct32_inv(z, threads=8))
    finally:
        for k, v in saved.items():
            codec.set_option(k, v)


@fuzz(25)
@given(ttype=st.integers(0, 1), size=st.sampled_from([4, 8, 16]), n=st.integers(1, 3000), kind=st.integers(0, 1), seed=st.integers(1, 1 << 30),
       tpb=st.sampled_from([64, 128, 192, 256]), per_wave=st.integers(1, 4))
def test_transform_set_random(codec, oracle, ttype, size, n, kind, seed, tpb, per_wave):
    saved = {k: codec.get_option(k) for k in ("dct32_wg_threads", "dct32_inv_blocks_per_wave", "adaptive_per_wave")}
    try:
        codec.set_option("dct32_wg_threads", tpb)
        codec.set_option("dct32_inv_blocks_per_wave", per_wave)          # tiles per wave of the inverse family
        codec.set_option("adaptive_per_wave", seed & 1)
        x = _data(kind, n, size * size, seed)
        fwd = oracle.transform_fwd(ttype, size, x)
        assert np.array_equal(codec.transform_fwd(ttype, size, x), fwd)
        assert np.array_equal(codec.transform_inv(ttype, size, fwd), oracle.transform_inv(ttype, size, fwd))
    finally:
        for k, v in saved.items():
            codec.set_option(k, v)


@fuzz(15)
@given(n=st.integers(1, 300), seed=st.integers(1, 1 << 30))
def test_intra_random(codec, oracle, n, seed):
    refs = intra_refs_np(n, seed)
    modes = ((np.arange(n) * 13 + seed) % 35).astype(np.uint8)
    pred = codec.intra32_predict(refs, modes)
    assert np.array_equal(pred, oracle.intra32_predict(refs, modes))
    m = min(n, 40)
    costs, best = codec.intra32_costs(refs[:m], pred[:m])
    ocosts, obest = oracle.intra32_costs(refs[:m], pred[:m])
    assert np.array_equal(costs, ocosts) and np.array_equal(best, obest)
    assert np.all(costs[np.arange(m), modes[:m]] == 0)        # the block predicted by mode k costs nothing under mode k


@fuzz(15)
@given(n=st.integers(1, 200), n_sets=st.integers(1, 40), seed=st.integers(1, 1 << 30), indexed=st.booleans())
def test_intra_residual_dct32_random(codec, oracle, n, n_sets, seed, indexed):
    """xIntra32ResidualDct32Dev on random counts, modes, shared reference sets and sources against oracle predictor -> residual -> pinned transform"""
    rs = np.random.RandomState(seed % (1 << 31))
    refs = intra_refs_np(n_sets if indexed else n, seed)
    modes = rs.randint(0, 35, n).astype(np.uint8)
    idx = rs.randint(0, n_sets, n).astype(np.uint32) if indexed else None
    src = rs.randint(0, 256, (n, 1024)).astype(np.uint8)
    src[rs.randint(0, n)] = rs.choice([0, 255])                         # a flat extreme block: the residual's +-255 edge
    got = codec.intra32_residual_dct32(refs, modes, src, idx)
    pred = oracle.intra32_predict(refs, modes, idx)
    assert np.array_equal(got, oracle.dct32_fwd(src.astype(np.int16) - pred.astype(np.int16)))


@fuzz(30)
@given(wb=st.integers(1, 14), hb=st.integers(1, 9), rng=st.integers(1, 24), tile_rows=st.sampled_from([0, 1, 2, 4, 8]),
       metric=st.sampled_from(["satd", "sad"]), seed=st.integers(1, 1 << 20))
def test_motion_search_random(codec, oracle, wb, hb, rng, tile_rows, metric, seed):
    from _util import me_frames
    w, h, pad = 8 * wb, 8 * hb, rng + (seed % 4)
    cur, refp = me_frames(w, h, pad, seed, mv=(min(rng, seed % 5), -min(rng, seed % 3)))
    saved = {k: codec.get_option(k) for k in ("me_tile_rows",)}
    try:
        codec.set_option("me_tile_rows", tile_rows)
        mv, cost, costs = codec.satd_search(cur, refp, pad, rng, want_costs=True, metric=metric)
        mv2, cost2, _ = codec.satd_search(cur, refp, pad, rng, metric=metric)
    finally:
        for k, v in saved.items():
            codec.set_option(k, v)
    omv, ocost, ocosts = oracle.satd_search(cur, refp, pad, rng, threads=8, want_costs=True, metric=metric)
    assert np.array_equal(costs, ocosts) and np.array_equal(cost, ocost) and np.array_equal(mv, omv)
    assert np.array_equal(cost2, ocost) and np.array_equal(mv2, omv)


@fuzz(30)
@given(bw=st.integers(1, 20), bh=st.integers(1, 20), rng=st.integers(1, 64), seed=st.integers(1, 1 << 20),
       tile_rows=st.sampled_from([1, 2, 4, 8]))
def test_motion_search_random_frames_ranges_and_shapes(codec, oracle, bw, bh, rng, seed, tile_rows):
    """Every window remainder (2R+1 = 8F + rem, rem in {1, 3, 5, 7}, F from 0 to 16), ragged tiles, all tile shapes:
    winners and costs of the SATD and SAD searches equal the oracle's brute force."""
    from _util import me_frames
    w, h = 8 * bw, 8 * bh
    if bw * bh * (2 * rng + 1) ** 2 > 1_500_000:                      # keep the brute-force oracle within seconds
        rng = max(1, rng // 4)
    pad = rng + (seed & 3)
    cur, refp = me_frames(w, h, pad, seed, mv=(min(rng, 2), -min(rng, 1)), noise=3)
    saved = {k: codec.get_option(k) for k in ("me_tile_rows",)}
    try:
        codec.set_option("me_tile_rows", tile_rows)
        mv, cost, _ = codec.satd_search(cur, refp, pad, rng)
        smv, scost, _ = codec.satd_search(cur, refp, pad, rng, metric="sad")
    finally:
        for k, v in saved.items():
            codec.set_option(k, v)
    omv, ocost, _ = oracle.satd_search(cur, refp, pad, rng, threads=8)
    assert np.array_equal(cost, ocost) and np.array_equal(mv, omv), (w, h, rng, tile_rows)
    omv, ocost, _ = oracle.satd_search(cur, refp, pad, rng, threads=8, metric="sad")
    assert np.array_equal(scost, ocost) and np.array_equal(smv, omv), (w, h, rng, tile_rows)


@fuzz(25)
@given(n_tiles=st.integers(1, 300), seed=st.integers(1, 1 << 20), inverse=st.integers(0, 1), tpw=st.integers(0, 5),
       tpb=st.sampled_from([64, 128, 256]), kind=st.integers(0, 2))
def test_mixed_class_tiles_random(codec, oracle, n_tiles, seed, inverse, tpw, tpb, kind):
    """xTransformTilesDev: random class per tile (all 13 valid classes), random tiles per wave / workgroup size, three data
    mixes -- equal to the oracle's per-class transforms tile by tile."""
    rs = np.random.RandomState(seed)
    valid = [t * 4 + l for t in range(4) for l in range(3)] + [3]       # (type, size 4/8/16) + (DCT-II, 32)
    cls = np.array([valid[i] for i in rs.randint(0, len(valid), n_tiles)], np.uint8)
    x = _data(kind, n_tiles, 1024, seed)
    if inverse and kind != 0:
        x = (x >> 2).astype(np.int16)                                    # keep coefficients in a sane range (clipping is still exercised by kind 2)
    saved = {k: codec.get_option(k) for k in ("tile_tiles_per_wave", "dct32_wg_threads")}
    try:
        codec.set_option("tile_tiles_per_wave", tpw)
        codec.set_option("dct32_wg_threads", tpb)
        din, dout, dcls = codec.alloc(n_tiles * 2048), codec.alloc(n_tiles * 2048), codec.alloc(max(n_tiles, 16))
        din.upload(x)
        dcls.upload(cls)
        codec.transform_tiles_dev(inverse, din.ptr, dout.ptr, n_tiles, 0, dcls.ptr)
        codec.stream_sync()
        got = dout.download(np.int16, n_tiles * 1024).reshape(n_tiles, 1024)
    finally:
        for k, v in saved.items():
            codec.set_option(k, v)
    for t in range(n_tiles):
        ttype, n = int(cls[t]) >> 2, (4, 8, 16, 32)[int(cls[t]) & 3]
        fn = oracle.transform_inv if inverse else oracle.transform_fwd
        assert np.array_equal(got[t], fn(ttype, n, x[t].reshape(-1, n * n)).ravel()), (t, ttype, n)


@fuzz(25)
@given(n=st.integers(1, 400), shift=st.integers(1, 15), kind=st.integers(0, 2), seed=st.integers(1, 1 << 30))
def test_one_dimensional_pass_random(codec, oracle, n, shift, kind, seed):
    """xDct32PassDev (round 3): random counts, every legal shift, three data mixes -- partialButterfly32 (src_tb/dct32.c:66-170)
    block by block on a sample, and two passes at the reference's shifts = the 2-D transform on all of them."""
    x = _data(kind, n, 1024, seed)
    got = codec.dct32_pass(x, shift)
    rs = np.random.RandomState(seed & 0xFFFF)
    for b in sorted(set([0, n - 1] + [int(v) for v in rs.randint(0, n, 4)])):
        assert np.array_equal(got[b], oracle.dct32_pass(x[b], shift)), (b, shift)
    assert np.array_equal(codec.dct32_pass(codec.dct32_pass(x, 4), 11), oracle.dct32_fwd(x))


@fuzz(25)
@given(n_dct=st.integers(0, 700), n_satd=st.integers(0, 9000), kind=st.integers(0, 2), seed=st.integers(1, 1 << 30),
       gpw=st.integers(0, 9))
def test_frame_lanes_random(codec, oracle, n_dct, n_satd, kind, seed, gpw):
    """xDct32SatdFrameDev (round 3): both lanes of a frame in one grid, random and ragged counts on either side (zero included),
    random SATD run length per wave -- equal to the oracle's two transforms."""
    x, d = _data(kind, max(n_dct, 1), 1024, seed), _data(kind, max(n_satd, 1), 64, seed + 1)
    saved = {k: codec.get_option(k) for k in ("satd_groups_per_wave",)}
    try:
        codec.set_option("satd_groups_per_wave", gpw)
        din, dout = codec.alloc(x.nbytes), codec.alloc(x.nbytes)
        sin, sout = codec.alloc(d.nbytes), codec.alloc(max(n_satd, 4) * 4)
        din.upload(x)
        sin.upload(d)
        codec.frame_lanes_dev(din.ptr, dout.ptr, n_dct, sin.ptr, sout.ptr, n_satd)
        codec.stream_sync()
        got_d = dout.download(np.int16, max(n_dct, 1) * 1024).reshape(-1, 1024)[:n_dct]
        got_s = sout.download(np.uint32, max(n_satd, 1))[:n_satd]
    finally:
        for k, v in saved.items():
            codec.set_option(k, v)
    if n_dct:
        assert np.array_equal(got_d, oracle.dct32_fwd(x[:n_dct]))
    if n_satd:
        assert np.array_equal(got_s, oracle.satd8x8(d[:n_satd]))


@fuzz(15)
@given(slot=st.integers(0, 1), size=st.sampled_from([4, 8, 16]), n_tiles=st.integers(1, 120), seed=st.integers(1, 1 << 20),
       inverse=st.integers(0, 1), extreme=st.booleans())
def test_installed_matrices_through_the_tile_launch(oracle, slot, size, n_tiles, seed, inverse, extreme):
    """xTransformSetMatrix + xTransformTilesDev (round 3: the tile kernel builds its operand images from the compact matrix
    table in LDS): a random int8 matrix in one slot, tiles of all four type codes at that size mixed with DCT-II 32 tiles --
    every tile equal to the oracle's generic-matrix transform with the slot's matrices."""
    cd = x266_amd.Codec(0)                                              # own context: the installed matrix must not leak into other tests
    rs = np.random.RandomState(seed)
    m = rs.randint(-128, 128, (size, size)).astype(np.int8)
    if extreme:
        m[rs.randint(0, size), :] = 127
        m[:, rs.randint(0, size)] = -128
    cd.set_transform_matrix(slot, size, m)
    mats = [cd.get_transform_matrix(s, size) for s in (0, 1)]
    l = {4: 0, 8: 1, 16: 2}[size]
    types = rs.randint(0, 4, n_tiles)
    cls = (types * 4 + l).astype(np.uint8)
    big = rs.rand(n_tiles) < 0.2
    cls[big] = 3
    x = _data(1 if extreme else 0, n_tiles, 1024, seed)
    if inverse:
        x = (x >> 3).astype(np.int16)
    din, dout, dcls = cd.alloc(n_tiles * 2048), cd.alloc(n_tiles * 2048), cd.alloc(max(n_tiles, 16))
    din.upload(x)
    dcls.upload(cls)
    cd.transform_tiles_dev(inverse, din.ptr, dout.ptr, n_tiles, 0, dcls.ptr)
    cd.stream_sync()
    got = dout.download(np.int16, n_tiles * 1024).reshape(n_tiles, 1024)
    for t in range(n_tiles):
        if big[t]:
            want = (oracle.dct32_inv if inverse else oracle.dct32_fwd)(x[t:t + 1])
        else:
            ty = int(types[t])
            mh, mv = mats[1 if ty in (1, 2) else 0], mats[1 if ty in (1, 3) else 0]
            want = oracle.transform_matrix_passes(mh, mv, x[t].reshape(-1, size * size), inverse=bool(inverse))
        assert np.array_equal(got[t], np.asarray(want).ravel()), (t, int(cls[t]))
    cd.close()


@fuzz(25)
@given(edge=st.sampled_from([4, 8, 16, 32, 64]), n=st.integers(0, 20000), seed=st.integers(1, 1 << 30), extreme=st.integers(0, 3))
def test_sad_batches_random(codec, edge, n, seed, extreme):
    """batched SAD (f3): every wave takes 256 chunks of 16 bytes, so counts around multiples of 256 / (edge*edge/16) blocks and the
    ragged last wave are what random sizes probe; extreme 1/2 = all-0 against all-255 (the largest sums), 3 = equal inputs"""
    if edge >= 32:
        n = n % 3000
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (n, edge * edge), dtype=np.uint8)
    b = rng.integers(0, 256, (n, edge * edge), dtype=np.uint8)
    if extreme == 1:
        a[:], b[:] = 0, 255
    elif extreme == 2:
        a[:], b[:] = 255, 0
    elif extreme == 3:
        b = a.copy()
    want = np.abs(a.astype(np.int32) - b.astype(np.int32)).sum(axis=1).astype(np.uint32)
    assert np.array_equal(codec.sad(edge, a, b), want)


@fuzz(25)
@given(tw=st.integers(1, 21), th=st.integers(1, 9), ctu=st.integers(0, 1), pitch=st.integers(1, 3), extreme=st.integers(0, 1), seed=st.integers(1, 1 << 20),
       wg=st.sampled_from([0, 64, 128, 256]), satd_variant=st.sampled_from([0, 1, 3]))
def test_tile_stage_random_frames(codec, oracle, tw, th, ctu, pitch, extreme, seed, wg, satd_variant):
    """The tile stage on random frames (any bytes are a valid tile array): luma and chroma residuals in both orders, planar and pitched chroma
    outputs, the fused tiles -> DCT32 / SATD forms for both, and the whole-CTU launch -- against the oracle's residuals run through the pinned
    kernels' oracles.  ctu = 1 rounds the frame to whole 64x64 CTUs (the 32x32 orders need them), else any multiple of 16."""
    w, h = (64 * ((tw + 3) // 4), 64 * ((th + 3) // 4)) if ctu else (16 * tw, 16 * th)
    nt = (w // 16) * (h // 16)
    rs = np.random.RandomState(seed)
    mk = (lambda: rs.choice(np.array([0, 255], np.uint8), nt * 512)) if extreme else (lambda: rs.randint(0, 256, nt * 512).astype(np.uint8))
    tc, tp = mk(), mk()
    dc, dp = codec.alloc(nt * 512), codec.alloc(nt * 512)
    dc.upload(tc); dp.upload(tp)
    saved = {k: codec.get_option(k) for k in ("dct32_wg_threads", "satd_variant")}
    try:
        codec.set_option("dct32_wg_threads", wg); codec.set_option("satd_variant", satd_variant)
        npl = w * h // 4
        for edge in ((8, 32) if ctu else (8,)):
            n = npl // (edge * edge)
            ou, ov = oracle.residual_chroma(tc, tp, w, h, edge)
            buf = codec.alloc(npl * 2 * pitch * 2 + 64)
            buf.upload(np.full(npl * pitch * 2 + 32, 0x1234, np.int16))
            vp = buf.ptr + (edge * edge * 2 if pitch > 1 else n * pitch * edge * edge * 2)
            codec.residual_chroma_dev(dc.ptr, dp.ptr, w, h, edge, buf.ptr, vp, pitch)
            codec.stream_sync()
            got = buf.download(np.int16, npl * pitch * 2 + 32)
            gu = got[: n * pitch * edge * edge].reshape(n, pitch, edge * edge)[:, 0]
            off = edge * edge if pitch > 1 else n * pitch * edge * edge
            gv = got[off: off + n * pitch * edge * edge - (edge * edge * (pitch - 1) if pitch > 1 else 0)]
            gv = np.concatenate([gv, np.zeros((-len(gv)) % (pitch * edge * edge), np.int16)]).reshape(n, pitch, edge * edge)[:, 0]
            assert np.array_equal(gu, ou.reshape(n, -1)) and np.array_equal(gv, ov.reshape(n, -1)), (edge, pitch)
            lum = oracle.residual_luma(tc, tp, w, h, edge)
            dl = codec.alloc(w * h * 2)
            codec.residual_luma_dev(dc.ptr, dp.ptr, w, h, edge, dl.ptr)
            codec.stream_sync()
            assert np.array_equal(dl.download(np.int16, w * h), lum), edge
            if edge == 8:
                cu, cv, cl = codec.alloc(n * 4), codec.alloc(n * 4), codec.alloc(w * h // 64 * 4)
                codec.satd8x8_chroma_from_tiles_dev(dc.ptr, dp.ptr, w, h, cu.ptr, cv.ptr)
                codec.satd8x8_from_tiles_dev(dc.ptr, dp.ptr, w, h, cl.ptr)
                codec.stream_sync()
                assert np.array_equal(cu.download(np.uint32, n), oracle.satd8x8(ou)) and np.array_equal(cv.download(np.uint32, n), oracle.satd8x8(ov))
                assert np.array_equal(cl.download(np.uint32, w * h // 64), oracle.satd8x8(lum))
            else:
                zu, zv, zl, zc = codec.alloc(n * 2048), codec.alloc(n * 2048), codec.alloc(w * h * 2), codec.alloc(n * 6 * 2048)
                codec.dct32_fwd_chroma_from_tiles_dev(dc.ptr, dp.ptr, w, h, zu.ptr, zv.ptr)
                codec.dct32_fwd_from_tiles_dev(dc.ptr, dp.ptr, w, h, zl.ptr)
                codec.dct32_fwd_ctu_from_tiles_dev(dc.ptr, dp.ptr, w, h, zc.ptr)
                codec.stream_sync()
                wu, wv, wl = oracle.dct32_fwd(ou), oracle.dct32_fwd(ov), oracle.dct32_fwd(lum)
                assert np.array_equal(zu.download(np.int16, n * 1024).reshape(n, 1024), wu) and np.array_equal(zv.download(np.int16, n * 1024).reshape(n, 1024), wv)
                assert np.array_equal(zl.download(np.int16, w * h).reshape(-1, 1024), wl)
                ctus = zc.download(np.int16, n * 6 * 1024).reshape(n, 6, 1024)
                wl_ctu = wl.reshape(h // 64, 2, w // 64, 2, 1024).transpose(0, 2, 1, 3, 4).reshape(n, 4, 1024)
                assert np.array_equal(ctus[:, :4], wl_ctu) and np.array_equal(ctus[:, 4], wu) and np.array_equal(ctus[:, 5], wv)
    finally:
        for k, v in saved.items():
            codec.set_option(k, v)
