"""CPU: structural properties of the oracle (table symmetries, dense == butterfly,
the self-defined inverse against a numpy statement of its definition, the
round-trip bound, SATD invariances, the PRNG twins)."""
import numpy as np

from _util import extremes_np, fullrange_np, residual_np, splitmix64


def test_table_structure(oracle):
    g = oracle.table().astype(np.int64)
    assert g.shape == (32, 32) and np.all(g[0] == 64)
    n = np.arange(32)
    for k in range(32):                         # g[k][31-n] = (-1)^k g[k][n]   
        assert np.array_equal(g[k, 31 - n], (-1) ** k * g[k, n])
    assert g.sum(axis=1).tolist() == [2048] + [0] * 31       # row sums (byte-plane offset fix relies on it)
    assert np.abs(g).max() == 90                             # fits int8
    # near-orthogonality of the integer basis
    gram = g @ g.T
    assert np.abs(gram - np.diag(np.diag(gram))).max() < 0.004 * np.diag(gram).min()
    assert np.all(np.abs(np.diag(gram) - 64 * 64 * 32) < 200)
    # column 0 reproduces the 32 magnitudes the generators start from
    assert g[:, 0].tolist() == [64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57,
                                54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4]


def test_dense_equals_butterfly(oracle):
    x = np.concatenate([residual_np(64 * 1024, 3), fullrange_np(64 * 1024, 4), extremes_np(64 * 1024, 5)])
    for blk in x.reshape(-1, 1024):
        for shift in (4, 11):
            assert np.array_equal(oracle.dct32_pass(blk, shift), oracle.dct32_pass(blk, shift, dense=True))


def test_fwd_is_two_passes_and_matches_numpy(oracle):
    g = oracle.table().astype(np.int64)
    x = fullrange_np(8 * 1024, 7).reshape(8, 32, 32)
    out = oracle.dct32_fwd(x)
    for b in range(8):
        y = ((np.einsum("kn,jn->kj", g, x[b].astype(np.int64)) + 8) >> 4).astype(np.int16)          # coef[k][j]
        z = ((np.einsum("vj,kj->vk", g, y.astype(np.int64)) + 1024) >> 11).astype(np.int16)          # dct[v][k]
        assert np.array_equal(out[b].reshape(32, 32), z)


def test_inverse_definition(oracle):
    """UNPINNED path: the C inverse equals this numpy statement of its definition."""
    g = oracle.table().astype(np.int64)
    z = np.concatenate([oracle.dct32_fwd(residual_np(6 * 1024, 9)).ravel(), fullrange_np(6 * 1024, 10)])
    z = z.reshape(12, 32, 32)
    out = oracle.dct32_inv(z)
    for b in range(12):
        t = np.clip((np.einsum("vy,vu->uy", g, z[b].astype(np.int64)) + 64) >> 7, -32768, 32767)    # T[u][y]
        r = np.clip((np.einsum("ux,uy->yx", g, t) + 2048) >> 12, -32768, 32767)                      # R[y][x]
        assert np.array_equal(out[b].reshape(32, 32), r.astype(np.int16))


def test_roundtrip_bound(oracle):
    """fwd -> inv of 9-bit residuals reconstructs within a few LSB (frozen bound)."""
    x = residual_np(4096 * 1024, 0x266)
    r = oracle.dct32_inv(oracle.dct32_fwd(x, threads=4), threads=4)
    err = np.abs(r.ravel().astype(np.int32) - x.astype(np.int32))
    assert err.max() <= 6
    assert err.mean() < 1.0


def test_satd_invariances(oracle):
    d = residual_np(512 * 64, 12).reshape(512, 8, 8)
    base = oracle.satd8x8(d)
    assert np.array_equal(oracle.satd8x8(-d), base)                          # |.| is even
    assert np.array_equal(oracle.satd8x8(d.transpose(0, 2, 1)), base)        # H X H^T symmetric in the two axes
    assert np.array_equal(oracle.satd8x8(d[:, ::-1, :]), base)               # row reversal = sign flips of H rows
    dc = np.full((1, 8, 8), 3, np.int16)
    assert oracle.satd8x8(dc)[0] == (3 * 64 + 2) >> 2


def test_satd_matches_numpy_truncation_model(oracle):
    """exact integer Hadamard, low 16 bits as signed, abs, sum."""
    h2 = np.array([[1, 1], [1, -1]], np.int64)
    h8 = np.kron(np.kron(h2, h2), h2)
    d = np.concatenate([fullrange_np(300 * 64, 13), extremes_np(300 * 64, 14)]).reshape(600, 8, 8)
    c = np.einsum("ij,bjk,lk->bil", h8, d.astype(np.int64), h8)
    c16 = ((c + 32768) % 65536) - 32768
    want = ((np.abs(c16).sum(axis=(1, 2)) + 2) >> 2).astype(np.uint32)
    assert np.array_equal(oracle.satd8x8(d), want)


def test_prng_twins(oracle):
    assert np.array_equal(oracle.fill_residual(5000, 0x266, 77), residual_np(5000, 0x266, 77))
    r = residual_np(200000, 1)
    assert r.min() >= -255 and r.max() <= 255 and abs(float(r.mean())) < 1.0
    assert splitmix64(0, 0, 1)[0] == np.uint64(0xE220A8397B1DCDAF)           # SplitMix64 reference value


# ---- mixed transform set (BASELINE configs[3]): DCT-II + closed-form DST-VII ---------------------------------------------
def test_transform_set_matrices(oracle):
    g = oracle.table()
    for n in (4, 8, 16, 32):
        m = oracle.transform_matrix(0, n)
        assert np.array_equal(m, g[:: 32 // n, :n])                    # sub-matrices of g_t32
    assert oracle.transform_matrix(1, 4).tolist() == [[29, 55, 74, 84], [74, 74, 0, -74], [84, -29, -74, 55],
                                                      [55, -84, 74, -29]]      # the VVC DST-VII 4-point table
    for n in (4, 8, 16):
        k, c = np.arange(n)[:, None], np.arange(n)[None, :]
        want = np.round(64 * np.sqrt(n) * np.sqrt(4.0 / (2 * n + 1)) * np.sin(np.pi * (2 * k + 1) * (c + 1) / (2 * n + 1)))
        m = oracle.transform_matrix(1, n)
        assert np.array_equal(m, want.astype(np.int16))
        assert np.abs(m).max() <= 90                                   # fits the int8 matrix core
        gram = m.astype(np.int64) @ m.T.astype(np.int64)
        assert np.abs(gram - np.diag(np.diag(gram))).max() < 0.01 * np.diag(gram).min()


def test_transform_set_generic_equals_pinned_dct32(oracle):
    x = np.concatenate([residual_np(40 * 1024, 31), fullrange_np(40 * 1024, 32)]).reshape(-1, 1024)
    assert np.array_equal(oracle.transform_fwd(0, 32, x), oracle.dct32_fwd(x))
    # the caller-supplied-matrix passes (what xTransformSetMatrix installs) with g_t32 are the pinned transform too,
    # forward and inverse, and with arbitrary int8 matrices they are the numpy statement
    g = oracle.table()
    assert np.array_equal(oracle.transform_matrix_passes(g, g, x), oracle.dct32_fwd(x))
    z = oracle.dct32_fwd(x)
    assert np.array_equal(oracle.transform_matrix_passes(g, g, z, inverse=True), oracle.dct32_inv(z))
    rng = np.random.default_rng(5)
    for n in (4, 8, 16):
        mh, mv = rng.integers(-128, 128, (n, n)), rng.integers(-128, 128, (n, n))
        s1, s2 = int(np.log2(n)) - 1, int(np.log2(n)) + 6
        xb = fullrange_np(5 * n * n, 70 + n).reshape(5, n, n)
        out = oracle.transform_matrix_passes(mh, mv, xb).reshape(5, n, n)
        inv = oracle.transform_matrix_passes(mh, mv, xb, inverse=True).reshape(5, n, n)
        for b in range(5):
            y = ((np.einsum("kc,jc->kj", mh, xb[b].astype(np.int64)) + (1 << (s1 - 1))) >> s1).astype(np.int16)
            assert np.array_equal(out[b], ((np.einsum("vj,kj->vk", mv, y.astype(np.int64)) + (1 << (s2 - 1))) >> s2).astype(np.int16))
            t = np.clip((np.einsum("vy,vu->uy", mv, xb[b].astype(np.int64)) + 64) >> 7, -32768, 32767)        # columns first
            assert np.array_equal(inv[b], np.clip((np.einsum("ux,uy->yx", mh, t) + 2048) >> 12, -32768, 32767).astype(np.int16))


def test_transform_set_matches_numpy(oracle):
    for ttype in (0, 1):
        for n in (4, 8, 16):
            m = oracle.transform_matrix(ttype, n).astype(np.int64)
            s1, s2 = int(np.log2(n)) - 1, int(np.log2(n)) + 6
            x = fullrange_np(6 * n * n, 40 + n).reshape(6, n, n)
            out = oracle.transform_fwd(ttype, n, x).reshape(6, n, n)
            for b in range(6):
                y = ((np.einsum("kc,jc->kj", m, x[b].astype(np.int64)) + (1 << (s1 - 1))) >> s1).astype(np.int16)
                z = ((np.einsum("vj,kj->vk", m, y.astype(np.int64)) + (1 << (s2 - 1))) >> s2).astype(np.int16)
                assert np.array_equal(out[b], z)


def test_transform_set_mixed_directions(oracle):
    """Types 2 and 3: DST-VII along rows with DCT-II vertically, and the other way round -- forward and inverse
    against a numpy statement with two matrices; the pure types are the special case of equal matrices."""
    for ttype, (th, tv) in ((2, (1, 0)), (3, (0, 1)), (0, (0, 0)), (1, (1, 1))):
        for n in (4, 8, 16):
            mh = oracle.transform_matrix(th, n).astype(np.int64)
            mv = oracle.transform_matrix(tv, n).astype(np.int64)
            s1, s2 = int(np.log2(n)) - 1, int(np.log2(n)) + 6
            x = fullrange_np(5 * n * n, 140 + n + ttype).reshape(5, n, n)
            out = oracle.transform_fwd(ttype, n, x).reshape(5, n, n)
            for b in range(5):
                y = ((np.einsum("kc,jc->kj", mh, x[b].astype(np.int64)) + (1 << (s1 - 1))) >> s1).astype(np.int16)   # rows: horizontal
                z = ((np.einsum("vj,kj->vk", mv, y.astype(np.int64)) + (1 << (s2 - 1))) >> s2).astype(np.int16)      # then vertical
                assert np.array_equal(out[b], z), (ttype, n)
            zz = fullrange_np(4 * n * n, 150 + n + ttype).reshape(4, n, n)
            inv = oracle.transform_inv(ttype, n, zz).reshape(4, n, n)
            for b in range(4):
                t = np.clip((np.einsum("kc,kj->jc", mv, zz[b].astype(np.int64)) + 64) >> 7, -32768, 32767)          # columns first: vertical
                r = np.clip((np.einsum("kc,kj->jc", mh, t) + 2048) >> 12, -32768, 32767)
                assert np.array_equal(inv[b], r.astype(np.int16)), (ttype, n)
            r9 = residual_np(300 * n * n, 160 + n).reshape(-1, n * n)
            rt = oracle.transform_inv(ttype, n, oracle.transform_fwd(ttype, n, r9))
            assert np.abs(rt.astype(np.int32) - r9.astype(np.int32)).max() <= 6
    assert oracle.lib.orc_transform_fwd(4, 8, None, None, 0) == -1


# ---- frame container (src/x266.cpp:56-63, 415-492) ---------------------------------------
def _yuv(w, h, seed):
    r = splitmix64(seed, 0, w * h * 3 // 2)
    b = (r & np.uint64(0xFF)).astype(np.uint8)
    return b[:w * h].reshape(h, w), b[w * h:w * h + w * h // 4].reshape(h // 2, w // 2), b[w * h + w * h // 4:].reshape(h // 2, w // 2)


def test_tile_layout_and_round_trip(oracle):
    w, h = 64, 48
    y, u, v = _yuv(w, h, 5)
    tiles = oracle.conv_input_fmt(y, u, v).reshape(h // 16, w // 16, 512)
    for ty in range(h // 16):
        for tx in range(w // 16):
            t = tiles[ty, tx]
            assert np.array_equal(t[:256].reshape(16, 16), y[16 * ty:16 * ty + 16, 16 * tx:16 * tx + 16])
            c = t[256:384].reshape(8, 8, 2)                                    # 8 rows of interleaved U,V pairs
            assert np.array_equal(c[:, :, 0], u[8 * ty:8 * ty + 8, 8 * tx:8 * tx + 8])
            assert np.array_equal(c[:, :, 1], v[8 * ty:8 * ty + 8, 8 * tx:8 * tx + 8])
            assert not t[384:].any()                                           # m_I is never written
    y2, u2, v2 = oracle.conv_output_420(tiles, w, h)
    assert np.array_equal(y2, y) and np.array_equal(u2, u) and np.array_equal(v2, v)


def test_residual_blocks_definition(oracle):
    w, h = 64, 64
    yc, uc, vc = _yuv(w, h, 6)
    yp, up, vp = _yuv(w, h, 7)
    tc, tp = oracle.conv_input_fmt(yc, uc, vc), oracle.conv_input_fmt(yp, up, vp)
    d = yc.astype(np.int16) - yp.astype(np.int16)
    for edge in (8, 32):
        want = d.reshape(h // edge, edge, w // edge, edge).transpose(0, 2, 1, 3).reshape(-1)
        assert np.array_equal(oracle.residual_luma(tc, tp, w, h, edge), want)


def test_chroma_residual_blocks_definition(oracle):
    """chroma of the tiled pair, by its definition on the PLANES the tiles were packed from (numpy): block b of the
    (w/2) x (h/2) plane, raster order; planar streams (pitch 1) and the CTU-ordered stream U0 V0 U1 V1 (pitch 2)"""
    for w, h in ((64, 64), (192, 128)):
        yc, uc, vc = _yuv(w, h, 16)
        yp, up, vp = _yuv(w, h, 17)
        tc, tp = oracle.conv_input_fmt(yc, uc, vc), oracle.conv_input_fmt(yp, up, vp)
        du, dv = uc.astype(np.int16) - up.astype(np.int16), vc.astype(np.int16) - vp.astype(np.int16)
        for edge in (8, 32):
            blocks = lambda d: d.reshape(h // 2 // edge, edge, w // 2 // edge, edge).transpose(0, 2, 1, 3).reshape(-1, edge * edge)
            ru, rv = oracle.residual_chroma(tc, tp, w, h, edge)
            assert np.array_equal(ru.reshape(-1, edge * edge), blocks(du)) and np.array_equal(rv.reshape(-1, edge * edge), blocks(dv))
            both, _ = oracle.residual_chroma(tc, tp, w, h, edge, block_pitch=2)
            both = both.reshape(-1, 2, edge * edge)
            assert np.array_equal(both[:, 0], blocks(du)) and np.array_equal(both[:, 1], blocks(dv))


def test_transform_set_inverse_definition(oracle):
    x = np.concatenate([residual_np(30 * 1024, 81), fullrange_np(30 * 1024, 82)]).reshape(-1, 1024)
    assert np.array_equal(oracle.transform_inv(0, 32, x), oracle.dct32_inv(x))            # overlaps the DCT32 inverse
    for ttype in (0, 1):
        for n in (4, 8, 16):
            m = oracle.transform_matrix(ttype, n).astype(np.int64)
            z = fullrange_np(5 * n * n, 90 + n).reshape(5, n, n)
            out = oracle.transform_inv(ttype, n, z).reshape(5, n, n)
            for b in range(5):
                t = np.clip((np.einsum("kc,kj->jc", m, z[b].astype(np.int64)) + 64) >> 7, -32768, 32767)
                r = np.clip((np.einsum("kc,kj->jc", m, t) + 2048) >> 12, -32768, 32767)
                assert np.array_equal(out[b], r.astype(np.int16))
            r9 = residual_np(400 * n * n, 95 + n).reshape(-1, n * n)
            rt = oracle.transform_inv(ttype, n, oracle.transform_fwd(ttype, n, r9))
            assert np.abs(rt.astype(np.int32) - r9.astype(np.int32)).max() <= 6


# ---- full-search harness (BASELINE configs[2]; UNPINNED upstream: order, tie-break, padding are this repo's) --------
def test_search_harness_is_ten_lines_of_numpy(oracle):
    """The harness stated independently of orc_satd8x8_search: for every 8x8 block and every displacement in
    [-R, R]^2, cost = satd8x8(cur - ref) with the pinned per-block function; candidates in raster order (dy ascending,
    then dx), the FIRST minimum wins; the reference is read at (x + dx, y + dy) of a frame padded by >= R."""
    from _util import me_frames
    for (w, h, rng, pad, seed) in ((24, 16, 3, 5, 1), (16, 24, 6, 6, 2), (8, 8, 2, 4, 3)):
        cur, refp = me_frames(w, h, pad, seed, mv=(1, -1), noise=2)
        if seed == 3:
            cur[:] = 7
            refp[:] = 9                                                  # all candidates tie: (-R, -R) must win
        mv, cost, costs = oracle.satd_search(cur, refp, pad, rng, want_costs=True)
        span = 2 * rng + 1
        for by in range(h // 8):
            for bx in range(w // 8):
                blk = cur[8 * by:8 * by + 8, 8 * bx:8 * bx + 8].astype(np.int16)
                cands = [(dy, dx) for dy in range(-rng, rng + 1) for dx in range(-rng, rng + 1)]          # raster order
                diffs = np.stack([blk - refp[pad + 8 * by + dy:pad + 8 * by + dy + 8, pad + 8 * bx + dx:pad + 8 * bx + dx + 8].astype(np.int16)
                                  for dy, dx in cands])
                c = oracle.satd8x8(diffs.reshape(-1, 64))                # the pinned cost (src_tb/satd.c via test_oracle_vs_ref)
                first = int(np.argmin(c))                                # np.argmin returns the FIRST minimum
                b = by * (w // 8) + bx
                assert np.array_equal(costs[b], c) and costs.shape[1] == span * span
                assert cost[b] == c[first] and tuple(mv[b]) == (cands[first][1], cands[first][0])


def test_tile_oracle_on_the_references_own_self_test_stimulus(oracle):
    """The reference's only statement about its container is a compiled-out self-test (src/x266.cpp:614-643): tmp[i] = i, two
    ref_block_t, xConvInputFmt(blocks, &tmp[0], &tmp[256], &tmp[320], 32, 32, 16), then xConvOutput420.  x266.cpp does not build here
    (MSVC-isms), so the tile oracle stays pinned by reading; this runs the restatement on exactly that call -- chroma planes inside
    the luma buffer included -- against a numpy statement of ref_block_t (src/x266.cpp:56-63) and checks the round trip."""
    import ctypes
    tmp = (np.arange(32 * 16 + 2 * 32 * 16 // 4) & 0xFF).astype(np.uint8)
    blocks = np.full(1024, 0xCD, np.uint8)
    P = ctypes.c_void_p
    oracle.lib.orc_conv_input_fmt(P(blocks.ctypes.data), P(tmp.ctypes.data), P(tmp.ctypes.data + 256), P(tmp.ctypes.data + 320), ctypes.c_ssize_t(32), 32, 16)
    y, u, v = tmp[:512].reshape(16, 32), tmp[256:384].reshape(8, 16), tmp[320:448].reshape(8, 16)
    for t in range(2):
        b = blocks[512 * t:512 * t + 512]
        assert np.array_equal(b[:256].reshape(16, 16), y[:, 16 * t:16 * t + 16])                       # m_Y[16*16]
        assert np.array_equal(b[256:384].reshape(8, 8, 2)[:, :, 0], u[:, 8 * t:8 * t + 8])             # m_C: U, V interleaved
        assert np.array_equal(b[256:384].reshape(8, 8, 2)[:, :, 1], v[:, 8 * t:8 * t + 8])
        assert np.all(b[384:] == 0xCD)                                                                 # m_I untouched
    oy, ou, ov = np.zeros(512, np.uint8), np.zeros(128, np.uint8), np.zeros(128, np.uint8)
    oracle.lib.orc_conv_output_420(P(blocks.ctypes.data), P(oy.ctypes.data), ctypes.c_ssize_t(32), P(ou.ctypes.data), P(ov.ctypes.data), ctypes.c_ssize_t(16), 32, 16)
    assert np.array_equal(oy, y.reshape(-1)) and np.array_equal(ou, u.reshape(-1)) and np.array_equal(ov, v.reshape(-1))
