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


@pytest.mark.parametrize("w,h", [(64, 64), (128, 192), (320, 64), (448, 832), (1920, 1088), (3136, 1792), (48, 16), (16, 16), (208, 112)])
def test_chroma_residual_then_transform_and_cost(codec, oracle, w, h):
    """The chroma half of the residual stage (m_C of ref_block_t, src/x266.cpp:56-63, 441-449): de-interleaved U / V int16 blocks in
    both orders and both output layouts, the chain into the transform / the cost, and the fused one-kernel forms -- vs the oracle
    and vs numpy on the planes the tiles were packed from.  Odd tile counts per row (ragged 8-tile groups) are among the sizes."""
    yc, uc, vc = _yuv(w, h, 31 + w)
    yp, up, vp = _yuv(w, h, 32 + w)
    dc, nt = _pack(codec, yc, uc, vc, w, h)
    dp, _ = _pack(codec, yp, up, vp, w, h)
    tc, tp = dc.download(np.uint8, nt * 512), dp.download(np.uint8, nt * 512)
    npl = (w // 2) * (h // 2)                                            # samples per chroma plane
    du_np, dv_np = uc.astype(np.int16) - up.astype(np.int16), vc.astype(np.int16) - vp.astype(np.int16)
    for edge in ((32, 8) if w % 64 == 0 and h % 64 == 0 else (8,)):
        n = npl // (edge * edge)
        blocks = lambda d: d.reshape(h // 2 // edge, edge, w // 2 // edge, edge).transpose(0, 2, 1, 3).reshape(-1, edge * edge)
        # planar streams
        dru, drv = codec.alloc(npl * 2), codec.alloc(npl * 2)
        codec.residual_chroma_dev(dc.ptr, dp.ptr, w, h, edge, dru.ptr, drv.ptr)
        codec.stream_sync()
        ru, rv = dru.download(np.int16, npl), drv.download(np.int16, npl)
        ou, ov = oracle.residual_chroma(tc, tp, w, h, edge)
        assert np.array_equal(ru, ou) and np.array_equal(rv, ov)
        assert np.array_equal(ru.reshape(n, -1), blocks(du_np)) and np.array_equal(rv.reshape(n, -1), blocks(dv_np))
        # CTU order in one buffer: U0 V0 U1 V1 ... ; holes of a wider pitch stay untouched
        for pitch in (2, 3):
            dboth = codec.alloc(npl * 2 * pitch)
            dboth.upload(np.full(npl * pitch, 0x7777, np.int16))
            codec.residual_chroma_dev(dc.ptr, dp.ptr, w, h, edge, dboth.ptr, dboth.ptr + edge * edge * 2, pitch)
            codec.stream_sync()
            both = dboth.download(np.int16, npl * pitch).reshape(n, pitch, edge * edge)
            assert np.array_equal(both[:, 0], blocks(du_np)) and np.array_equal(both[:, 1], blocks(dv_np))
            assert pitch == 2 or np.all(both[:, 2] == 0x7777)
        if edge == 32:
            want_u, want_v = oracle.dct32_fwd(ru, threads=8), oracle.dct32_fwd(rv, threads=8)
            dz = codec.alloc(npl * 2)                                  # chain: residual -> the pinned transform
            codec.dct32_fwd_dev(dru.ptr, dz.ptr, n)
            codec.stream_sync()
            assert np.array_equal(dz.download(np.int16, npl).reshape(n, 1024), want_u)
            dzu, dzv = codec.alloc(npl * 2), codec.alloc(npl * 2)       # fused: tiles -> both planes' coefficients in one kernel
            codec.dct32_fwd_chroma_from_tiles_dev(dc.ptr, dp.ptr, w, h, dzu.ptr, dzv.ptr)
            codec.stream_sync()
            assert np.array_equal(dzu.download(np.int16, npl).reshape(n, 1024), want_u)
            assert np.array_equal(dzv.download(np.int16, npl).reshape(n, 1024), want_v)
            dzb = codec.alloc(npl * 4)                                 # ... CTU-ordered
            codec.dct32_fwd_chroma_from_tiles_dev(dc.ptr, dp.ptr, w, h, dzb.ptr, dzb.ptr + 2048, 2)
            codec.stream_sync()
            zb = dzb.download(np.int16, npl * 2).reshape(n, 2, 1024)
            assert np.array_equal(zb[:, 0], want_u) and np.array_equal(zb[:, 1], want_v)
        else:
            want_u, want_v = oracle.satd8x8(ru, threads=8), oracle.satd8x8(rv, threads=8)
            ds = codec.alloc(n * 4)
            codec.satd8x8_dev(drv.ptr, ds.ptr, n)
            codec.stream_sync()
            assert np.array_equal(ds.download(np.uint32, n), want_v)
            dsu, dsv = codec.alloc(n * 4), codec.alloc(n * 4)
            codec.satd8x8_chroma_from_tiles_dev(dc.ptr, dp.ptr, w, h, dsu.ptr, dsv.ptr)
            codec.stream_sync()
            assert np.array_equal(dsu.download(np.uint32, n), want_u) and np.array_equal(dsv.download(np.uint32, n), want_v)
            dsb = codec.alloc(n * 8)
            codec.satd8x8_chroma_from_tiles_dev(dc.ptr, dp.ptr, w, h, dsb.ptr, dsb.ptr + 4, 2)
            codec.stream_sync()
            sb = dsb.download(np.uint32, 2 * n).reshape(n, 2)
            assert np.array_equal(sb[:, 0], want_u) and np.array_equal(sb[:, 1], want_v)


@pytest.mark.parametrize("w,h", [(64, 64), (192, 128), (448, 832), (1920, 1088 - 1088 % 64), (3136, 1792)])
def test_whole_ctu_in_one_launch(codec, oracle, w, h):
    """xDct32FwdCtuFromTilesDev: per 64x64 CTU the six coefficient blocks Y0 Y1 Y2 Y3 U V (12 KiB, CTU raster order) from ONE launch --
    against the oracle on the planes the tiles were packed from, and against the frame-raster luma / planar chroma calls it re-orders."""
    yc, uc, vc = _yuv(w, h, 41 + w)
    yp, up, vp = _yuv(w, h, 42 + w)
    dc, nt = _pack(codec, yc, uc, vc, w, h)
    dp, _ = _pack(codec, yp, up, vp, w, h)
    n_ctu = (w // 64) * (h // 64)
    dz = codec.alloc(n_ctu * 6 * 2048)
    dz.upload(np.full(n_ctu * 6 * 1024, 0x5555, np.int16))
    codec.dct32_fwd_ctu_from_tiles_dev(dc.ptr, dp.ptr, w, h, dz.ptr)
    codec.stream_sync()
    got = dz.download(np.int16, n_ctu * 6 * 1024).reshape(n_ctu, 6, 1024)
    dy = (yc.astype(np.int16) - yp.astype(np.int16)).reshape(h // 64, 2, 32, w // 64, 2, 32).transpose(0, 3, 1, 4, 2, 5).reshape(n_ctu, 4, 1024)
    du = (uc.astype(np.int16) - up.astype(np.int16)).reshape(h // 64, 32, w // 64, 32).transpose(0, 2, 1, 3).reshape(n_ctu, 1024)
    dv = (vc.astype(np.int16) - vp.astype(np.int16)).reshape(h // 64, 32, w // 64, 32).transpose(0, 2, 1, 3).reshape(n_ctu, 1024)
    want = oracle.dct32_fwd(np.concatenate([dy, du[:, None], dv[:, None]], axis=1).reshape(-1, 1024), threads=8).reshape(n_ctu, 6, 1024)
    assert np.array_equal(got, want)
    # the two calls it stands for: luma in frame raster order of 32x32 blocks, chroma per CTU
    dl, dcu, dcv = codec.alloc(w * h * 2), codec.alloc(n_ctu * 2048), codec.alloc(n_ctu * 2048)
    codec.dct32_fwd_from_tiles_dev(dc.ptr, dp.ptr, w, h, dl.ptr)
    codec.dct32_fwd_chroma_from_tiles_dev(dc.ptr, dp.ptr, w, h, dcu.ptr, dcv.ptr)
    codec.stream_sync()
    luma = dl.download(np.int16, w * h).reshape(h // 64, 2, w // 64, 2, 1024).transpose(0, 2, 1, 3, 4).reshape(n_ctu, 4, 1024)
    assert np.array_equal(got[:, :4], luma)
    assert np.array_equal(got[:, 4], dcu.download(np.int16, n_ctu * 1024).reshape(n_ctu, 1024))
    assert np.array_equal(got[:, 5], dcv.download(np.int16, n_ctu * 1024).reshape(n_ctu, 1024))
    assert codec.L.xDct32FwdCtuFromTilesDev(codec.ctx, dc.ptr, dp.ptr, w + 32, h, dz.ptr, None) < 0      # not whole CTUs


def test_chroma_extreme_pixels(codec, oracle):
    """0 / 255 chroma samples: differences of +-255 through the single-byte-plane fused paths"""
    w, h = 128, 64
    r = splitmix64(98, 0, w * h)
    pl = lambda k: np.where((r[:w * h // 4] >> np.uint64(k)) & np.uint64(1), 255, 0).astype(np.uint8).reshape(h // 2, w // 2)
    z = np.zeros((h, w), np.uint8)
    dc, nt = _pack(codec, z, pl(0), pl(1), w, h)
    dp, _ = _pack(codec, z, pl(2), pl(3), w, h)
    tc, tp = dc.download(np.uint8, nt * 512), dp.download(np.uint8, nt * 512)
    npl = w * h // 4
    dzu, dzv = codec.alloc(npl * 2), codec.alloc(npl * 2)
    codec.dct32_fwd_chroma_from_tiles_dev(dc.ptr, dp.ptr, w, h, dzu.ptr, dzv.ptr)
    codec.stream_sync()
    ou, ov = oracle.residual_chroma(tc, tp, w, h, 32)
    assert np.abs(ou).max() == 255
    assert np.array_equal(dzu.download(np.int16, npl), oracle.dct32_fwd(ou).ravel()) and np.array_equal(dzv.download(np.int16, npl), oracle.dct32_fwd(ov).ravel())
    dsu, dsv = codec.alloc(npl // 64 * 4), codec.alloc(npl // 64 * 4)
    codec.satd8x8_chroma_from_tiles_dev(dc.ptr, dp.ptr, w, h, dsu.ptr, dsv.ptr)
    codec.stream_sync()
    ou8, ov8 = oracle.residual_chroma(tc, tp, w, h, 8)
    assert np.array_equal(dsu.download(np.uint32, npl // 64), oracle.satd8x8(ou8)) and np.array_equal(dsv.download(np.uint32, npl // 64), oracle.satd8x8(ov8))


def test_chroma_argument_errors(codec):
    L = codec.L
    buf = codec.alloc(1 << 20)
    p = buf.ptr
    assert L.xResidualChromaDev(codec.ctx, p, p, 96, 64, 32, p + 65536, p + 131072, 1, None) < 0          # 96 % 64: a 32x32 chroma block is a CTU's
    assert L.xResidualChromaDev(codec.ctx, p, p, 64, 64, 16, p + 65536, p + 131072, 1, None) < 0          # edge
    assert L.xResidualChromaDev(codec.ctx, p, p, 64, 64, 8, p + 65536, p + 131072, 0, None) < 0           # pitch 0
    assert L.xResidualChromaDev(codec.ctx, p, p, 64, 64, 8, p + 65536, p + 65536 + 128, 1, None) < 0      # V inside U's stream
    assert L.xResidualChromaDev(codec.ctx, p, p, 64, 64, 8, p + 65536, p + 65536 + 256, 2, None) < 0      # V on U's next block
    assert L.xResidualChromaDev(codec.ctx, p, p, 64, 64, 8, p + 65536, p + 65536 + 128, 2, None) == 0     # the interleaved form
    assert L.xDct32FwdChromaFromTilesDev(codec.ctx, p, p, 64, 32, p + 65536, p + 131072, 1, None) < 0     # height % 64
    assert L.xDct32FwdChromaFromTilesDev(codec.ctx, p, p, 64, 64, p + 65536, None, 1, None) < 0
    assert L.xSatd8x8ChromaFromTilesDev(codec.ctx, p, p, 24, 16, p + 65536, p + 131072, 1, None) < 0      # width % 16
    assert L.xSatd8x8ChromaFromTilesDev(codec.ctx, p, p, 32, 16, p + 65536, p + 65536, 2, None) < 0       # same pointer twice
    codec.stream_sync()


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


def test_chroma_of_tile_frames_beyond_4_gib(codec, oracle):
    """The chroma kernels on a 65600 x 32768 frame (8 396 800 tiles = 4.3 GB per tile array): CTUs at the start, around the tile whose byte
    offset is 2^32 and at the frame's far corners -- residual in both orders, fused transform, fused cost, planar outputs."""
    w, h = 65536 + 64, 32768
    tiles_x, nt = w // 16, (w // 16) * (h // 16)
    assert nt * 512 > (1 << 32)
    d_cur, d_pred = codec.alloc(nt * 512), codec.alloc(nt * 512)
    codec.fill_residual_dev(d_cur.ptr, nt * 256, 0xC2)
    codec.fill_residual_dev(d_pred.ptr, nt * 256, 0xC3)
    npl = w * h // 4
    d_r32, d_r8, d_coef, d_cost = codec.alloc(npl * 4), codec.alloc(npl * 4), codec.alloc(npl * 4), codec.alloc(nt * 8)
    codec.residual_chroma_dev(d_cur.ptr, d_pred.ptr, w, h, 32, d_r32.ptr, d_r32.ptr + npl * 2)
    codec.residual_chroma_dev(d_cur.ptr, d_pred.ptr, w, h, 8, d_r8.ptr, d_r8.ptr + npl * 2)
    codec.dct32_fwd_chroma_from_tiles_dev(d_cur.ptr, d_pred.ptr, w, h, d_coef.ptr, d_coef.ptr + npl * 2)
    codec.satd8x8_chroma_from_tiles_dev(d_cur.ptr, d_pred.ptr, w, h, d_cost.ptr, d_cost.ptr + nt * 4)
    codec.stream_sync()

    def fetch(buf, byte_off, count, dtype):
        out = np.empty(count, dtype)
        codec._check(codec.L.xHipMemcpyD2H(codec.ctx, out.ctypes.data, buf.ptr + byte_off, out.nbytes), "D2H")
        return out

    t_edge = (1 << 32) // 512
    ctus_x = w // 64
    for cy, cx in [(0, 0), (t_edge // tiles_x // 4, (t_edge % tiles_x) // 4), (h // 64 - 1, ctus_x - 1), (h // 64 - 1, 0), (h // 128, ctus_x - 1)]:
        sixteen = lambda buf: np.concatenate([fetch(buf, ((4 * cy + j) * tiles_x + 4 * cx) * 512, 2048, np.uint8) for j in range(4)])   # the CTU's 4 x 4 tiles
        tc, tp = sixteen(d_cur), sixteen(d_pred)
        u32, v32 = oracle.residual_chroma(tc, tp, 64, 64, 32)
        u8, v8 = oracle.residual_chroma(tc, tp, 64, 64, 8)
        ctu = cy * ctus_x + cx
        for plane, (r32, r8) in enumerate(((u32, u8), (v32, v8))):
            base = plane * npl * 2
            assert np.array_equal(fetch(d_r32, base + ctu * 2048, 1024, np.int16), r32), (cy, cx, plane)
            assert np.array_equal(fetch(d_coef, base + ctu * 2048, 1024, np.int16), oracle.dct32_fwd(r32).ravel()), (cy, cx, plane)
            want_cost = oracle.satd8x8(r8.reshape(16, 64)).reshape(4, 4)
            for j in range(4):
                first = (4 * cy + j) * tiles_x + 4 * cx                  # four consecutive tiles = 8x8 chroma blocks of tile row 4 cy + j
                assert np.array_equal(fetch(d_r8, base + first * 128, 256, np.int16), r8.reshape(4, 4, 64)[j].ravel()), (cy, cx, plane, j)
                assert np.array_equal(fetch(d_cost, plane * nt * 4 + first * 4, 4, np.uint32), want_cost[j]), (cy, cx, plane, j)


@pytest.mark.parametrize("size", [(), ("256", "192"), ("3840", "2176")])
def test_plain_c_host_takes_a_420_frame_through_the_tile_stage(size):
    """host/frame420_example.c: planar YUV -> tiles -> whole-CTU coefficients + luma / chroma SATD costs from a C host without HIP headers,
    every fused output compared there with the two-step calls (residual in HBM, then the pinned batch kernels)."""
    import json
    import os
    import subprocess
    from _util import ROOT
    exe = os.path.join(ROOT, "host", "frame420_example")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "--no-print-directory"])
    out = subprocess.run([exe, *size], timeout=300, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["fused_equals_two_step"] is True and d["luma_satd_sum"] > 0 and d["chroma_satd_sum"] > 0
    w, h = (int(size[0]), int(size[1])) if size else (1920, 1088)
    assert d["ctus"] == (w // 64) * (h // 64) and d["coefficient_bytes"] == d["ctus"] * 12288
