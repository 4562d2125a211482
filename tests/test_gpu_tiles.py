"""GPU (-m gpu): frame container conversion (xConvInputFmt / xConvOutput420 of src/x266.cpp on
the device) and residual formation, against the oracle; plus the chain tiles -> residual ->
DCT32 / SATD that an encoder would run."""
import ctypes

import numpy as np
import pytest

from _util import splitmix64

pytestmark = pytest.mark.gpu


def _yuv(w, h, seed, strd=None):
    strd = strd or w
    r = splitmix64(seed, 0, strd * h * 3 // 2 + 64)
    b = (r & np.uint64(0xFF)).astype(np.uint8)
    y = b[:strd * h].reshape(h, strd)
    u = b[strd * h:strd * h + strd * h // 4].reshape(h // 2, strd // 2)
    v = b[strd * h + strd * h // 4:strd * h + strd * h // 2].reshape(h // 2, strd // 2)
    return y, u, v


def _pack(codec, y, u, v, w, h):
    dy, du, dv = codec.alloc(y.nbytes), codec.alloc(u.nbytes), codec.alloc(v.nbytes)
    dy.upload(y); du.upload(u); dv.upload(v)
    nt = (w // 16) * (h // 16)
    dt = codec.alloc(nt * 512)
    dt.upload(np.full(nt * 512, 0xEE, np.uint8))                      # m_I must survive untouched
    codec.conv_input_fmt_dev(dt.ptr, dy.ptr, du.ptr, dv.ptr, y.strides[0], w, h)
    codec.stream_sync()
    return dt, nt


@pytest.mark.parametrize("w,h,strd", [(16, 16, 16), (64, 48, 64), (320, 240, 320), (128, 64, 160), (3840, 2160, 3840)])
def test_pack_and_unpack(codec, oracle, w, h, strd):
    y, u, v = _yuv(w, h, 11 + w, strd)
    dt, nt = _pack(codec, y, u, v, w, h)
    tiles = dt.download(np.uint8, nt * 512).reshape(nt, 512)
    if strd == w:
        want = oracle.conv_input_fmt(y, u, v).reshape(nt, 512)
        assert np.array_equal(tiles[:, :384], want[:, :384])
    assert np.all(tiles[:, 384:] == 0xEE)
    # layout, independent of the oracle
    t0 = tiles[0]
    assert np.array_equal(t0[:256].reshape(16, 16), y[:16, :16])
    assert np.array_equal(t0[256:384].reshape(8, 8, 2)[:, :, 0], u[:8, :8]) and np.array_equal(t0[256:384].reshape(8, 8, 2)[:, :, 1], v[:8, :8])
    # unpack into fresh planes
    oy, ou, ov = codec.alloc(w * h), codec.alloc(w * h // 4), codec.alloc(w * h // 4)
    codec.conv_output_420_dev(dt.ptr, oy.ptr, w, ou.ptr, ov.ptr, w // 2, w, h)
    codec.stream_sync()
    assert np.array_equal(oy.download(np.uint8, w * h).reshape(h, w), y[:, :w])
    assert np.array_equal(ou.download(np.uint8, w * h // 4).reshape(h // 2, w // 2), u[:, :w // 2])
    assert np.array_equal(ov.download(np.uint8, w * h // 4).reshape(h // 2, w // 2), v[:, :w // 2])


@pytest.mark.parametrize("w,h", [(32, 32), (64, 96), (160, 96), (224, 416), (1920, 1088), (3104, 1760)])
def test_residual_then_transform_and_cost(codec, oracle, w, h):
    yc, uc, vc = _yuv(w, h, 21 + w)
    yp, up, vp = _yuv(w, h, 22 + w)
    dc, nt = _pack(codec, yc, uc, vc, w, h)
    dp, _ = _pack(codec, yp, up, vp, w, h)
    tc, tp = dc.download(np.uint8, nt * 512), dp.download(np.uint8, nt * 512)
    for edge in (32, 8):
        dres = codec.alloc(w * h * 2)
        codec.residual_luma_dev(dc.ptr, dp.ptr, w, h, edge, dres.ptr)
        codec.stream_sync()
        res = dres.download(np.int16, w * h)
        assert np.array_equal(res, oracle.residual_luma(tc, tp, w, h, edge))
        d = yc.astype(np.int16) - yp.astype(np.int16)
        assert np.array_equal(res, d.reshape(h // edge, edge, w // edge, edge).transpose(0, 2, 1, 3).reshape(-1))
        n = w * h // (edge * edge)
        if edge == 32:                                                 # ... straight into the transform
            dz = codec.alloc(w * h * 2)
            codec.dct32_fwd_dev(dres.ptr, dz.ptr, n)
            codec.stream_sync()
            want = oracle.dct32_fwd(res, threads=8)
            assert np.array_equal(dz.download(np.int16, w * h).reshape(n, 1024), want)
            dz2 = codec.alloc(w * h * 2)                               # fused: tiles -> coefficients in one kernel
            codec.dct32_fwd_from_tiles_dev(dc.ptr, dp.ptr, w, h, dz2.ptr)
            codec.stream_sync()
            assert np.array_equal(dz2.download(np.int16, w * h).reshape(n, 1024), want)
        else:                                                          # ... straight into the cost
            ds = codec.alloc(n * 4)
            codec.satd8x8_dev(dres.ptr, ds.ptr, n)
            codec.stream_sync()
            want = oracle.satd8x8(res, threads=8)
            assert np.array_equal(ds.download(np.uint32, n), want)
            for variant in (0, 1, 3):                                  # fused: tiles -> costs in one kernel; by frame size / staged body / LDS-DMA body
                codec.set_option("satd_variant", variant)
                ds2 = codec.alloc(n * 4)
                ds2.upload(np.full(n, 0xFFFFFFFF, np.uint32))
                codec.satd8x8_from_tiles_dev(dc.ptr, dp.ptr, w, h, ds2.ptr)
                codec.stream_sync()
                assert np.array_equal(ds2.download(np.uint32, n), want), variant
            codec.set_option("satd_variant", 0)


def test_fused_transform_extreme_pixels(codec, oracle):
    """0 / 255 pixels drive the differences to +-255, the edge of the single-byte-plane path."""
    w, h = 64, 64
    r = splitmix64(99, 0, 2 * w * h)
    yc = np.where(r[: w * h] & np.uint64(1), 255, 0).astype(np.uint8).reshape(h, w)
    yp = np.where(r[w * h:] & np.uint64(2), 255, 0).astype(np.uint8).reshape(h, w)
    z = np.zeros((h // 2, w // 2), np.uint8)
    dc, nt = _pack(codec, yc, z, z, w, h)
    dp, _ = _pack(codec, yp, z, z, w, h)
    dz = codec.alloc(w * h * 2)
    codec.dct32_fwd_from_tiles_dev(dc.ptr, dp.ptr, w, h, dz.ptr)
    codec.stream_sync()
    d = (yc.astype(np.int16) - yp.astype(np.int16)).reshape(h // 32, 32, w // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 1024)
    assert np.array_equal(dz.download(np.int16, w * h).reshape(-1, 1024), oracle.dct32_fwd(d))
    ds = codec.alloc(w * h // 64 * 4)
    codec.satd8x8_from_tiles_dev(dc.ptr, dp.ptr, w, h, ds.ptr)
    codec.stream_sync()
    d8 = (yc.astype(np.int16) - yp.astype(np.int16)).reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    assert np.array_equal(ds.download(np.uint32, w * h // 64), oracle.satd8x8(d8))


def test_argument_errors(codec):
    L = codec.L
    buf = codec.alloc(1 << 16)
    assert L.xDct32FwdFromTilesDev(codec.ctx, buf.ptr, buf.ptr, 48, 32, buf.ptr, None) < 0                 # 48 % 32
    assert L.xConvInputFmtDev(codec.ctx, buf.ptr, buf.ptr, buf.ptr, buf.ptr, 24, 24, 16, None) < 0         # width % 16
    assert L.xConvInputFmtDev(codec.ctx, buf.ptr, None, buf.ptr, buf.ptr, 32, 32, 16, None) < 0
    assert L.xResidualLumaDev(codec.ctx, buf.ptr, buf.ptr, 48, 32, 32, buf.ptr, None) < 0                  # 48 % 32
    assert L.xResidualLumaDev(codec.ctx, buf.ptr, buf.ptr, 32, 32, 16, buf.ptr, None) < 0                  # edge


def test_the_references_own_self_test_stimulus(codec, oracle):
    """src/x266.cpp:614-643 (`#if TEST_xConvInputFmt`, compiled out upstream): tmp[i] = i over 32*16 + 2*32*16/4 bytes, two ref_block_t,
    xConvInputFmt(blocks, &tmp[0], &tmp[256], &tmp[320], strd 32, 32 x 16) -- with the U and V planes deliberately INSIDE the luma
    buffer -- then xConvOutput420 back.  The same call on the device: tiles equal the oracle's and a numpy statement of the layout
    (ref_block_t, src/x266.cpp:56-63), the m_I part is not touched, and the round trip returns the planes."""
    tmp = (np.arange(32 * 16 + 2 * 32 * 16 // 4) & 0xFF).astype(np.uint8)
    d_tmp = codec.alloc(tmp.nbytes)
    d_tmp.upload(tmp)
    d_tiles = codec.alloc(2 * 512)
    d_tiles.upload(np.full(2 * 512, 0xCD, np.uint8))                     # memset(blocks, 0xCD, sizeof(blocks)), :625
    codec.conv_input_fmt_dev(d_tiles.ptr, d_tmp.ptr, d_tmp.ptr + 256, d_tmp.ptr + 320, 32, 32, 16)
    codec.stream_sync()
    tiles = d_tiles.download(np.uint8, 1024).reshape(2, 512)
    y = tmp[:512].reshape(16, 32)
    u = tmp[256:256 + 128].reshape(8, 16)
    v = tmp[320:320 + 128].reshape(8, 16)
    want = np.full((2, 512), 0xCD, np.uint8)
    for t in range(2):
        want[t, :256] = y[:, 16 * t:16 * t + 16].reshape(-1)
        want[t, 256:384] = np.stack([u[:, 8 * t:8 * t + 8], v[:, 8 * t:8 * t + 8]], axis=-1).reshape(-1)
    assert np.array_equal(tiles, want)
    orc = np.full(1024, 0xCD, np.uint8)
    oracle.lib.orc_conv_input_fmt(orc.ctypes.data_as(ctypes.c_void_p), tmp.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(tmp.ctypes.data + 256),
                                  ctypes.c_void_p(tmp.ctypes.data + 320), ctypes.c_ssize_t(32), 32, 16)
    assert np.array_equal(orc.reshape(2, 512)[:, :384], tiles[:, :384])
    oy, ou, ov = codec.alloc(512), codec.alloc(128), codec.alloc(128)
    codec.conv_output_420_dev(d_tiles.ptr, oy.ptr, 32, ou.ptr, ov.ptr, 16, 32, 16)
    codec.stream_sync()
    assert np.array_equal(oy.download(np.uint8, 512), y.reshape(-1))
    assert np.array_equal(ou.download(np.uint8, 128), u.reshape(-1)) and np.array_equal(ov.download(np.uint8, 128), v.reshape(-1))


def test_tile_frames_beyond_4_gib(codec, oracle):
    """Maximum sizes: a 65568 x 32768 frame = 8 392 704 tiles = 4.3 GB per tile array and per int16 output (and more than 2^31 pixels): 32x32 regions at the
    start, around the tile whose byte offset is 2^32 and at the frame's last corner -- residual in both orders, fused transform, fused cost (LDS-DMA body)."""
    w, h = 65536 + 32, 32768
    tiles_x, nt = w // 16, (w // 16) * (h // 16)
    assert nt * 512 > (1 << 32) and w * h > (1 << 31)
    d_cur, d_pred = codec.alloc(nt * 512), codec.alloc(nt * 512)
    codec.fill_residual_dev(d_cur.ptr, nt * 256, 0xC0)                 # any bytes are a valid tile array
    codec.fill_residual_dev(d_pred.ptr, nt * 256, 0xC1)
    d_res32, d_res8, d_coef, d_cost = codec.alloc(w * h * 2), codec.alloc(w * h * 2), codec.alloc(w * h * 2), codec.alloc(w * h // 64 * 4)
    codec.residual_luma_dev(d_cur.ptr, d_pred.ptr, w, h, 32, d_res32.ptr)
    codec.residual_luma_dev(d_cur.ptr, d_pred.ptr, w, h, 8, d_res8.ptr)
    codec.dct32_fwd_from_tiles_dev(d_cur.ptr, d_pred.ptr, w, h, d_coef.ptr)
    codec.satd8x8_from_tiles_dev(d_cur.ptr, d_pred.ptr, w, h, d_cost.ptr)
    codec.stream_sync()

    def fetch(buf, byte_off, count, dtype):
        out = np.empty(count, dtype)
        codec._check(codec.L.xHipMemcpyD2H(codec.ctx, out.ctypes.data, buf.ptr + byte_off, out.nbytes), "D2H")
        return out

    t_edge = (1 << 32) // 512                                          # the tile that starts at byte 2^32
    regions = [(0, 0), (t_edge // tiles_x // 2, (t_edge % tiles_x) // 2), (h // 32 - 1, w // 32 - 1), (h // 32 - 1, 0), (h // 64, w // 32 - 1)]
    for by, bx in regions:
        four = lambda buf: np.concatenate([fetch(buf, ((2 * by + j) * tiles_x + 2 * bx) * 512, 1024, np.uint8) for j in (0, 1)])   # the region's 2 x 2 tiles
        tc, tp = four(d_cur), four(d_pred)
        r32 = oracle.residual_luma(tc, tp, 32, 32, 32)
        blk = by * (w // 32) + bx
        assert np.array_equal(fetch(d_res32, blk * 2048, 1024, np.int16), r32), (by, bx)
        assert np.array_equal(fetch(d_coef, blk * 2048, 1024, np.int16), oracle.dct32_fwd(r32).ravel()), (by, bx)
        r8 = oracle.residual_luma(tc, tp, 32, 32, 8).reshape(4, 4, 64)
        want_cost = oracle.satd8x8(r8.reshape(16, 64)).reshape(4, 4)
        for j in range(4):
            first = (4 * by + j) * (w // 8) + 4 * bx                   # four consecutive 8x8 blocks of block row 4 by + j
            assert np.array_equal(fetch(d_res8, first * 128, 256, np.int16), r8[j].ravel()), (by, bx, j)
            assert np.array_equal(fetch(d_cost, first * 4, 4, np.uint32), want_cost[j]), (by, bx, j)
