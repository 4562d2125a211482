"""GPU (-m gpu): the mixed transform set of BASELINE configs[3] (DCT-II + closed-form DST-VII) through
xTransformFwdBatchDev, bit-exact against the oracle (DCT-II 4/8/16/32, DST-VII
4/8/16; contiguous batches and per-CTU mixed batches placed by offset tables)."""
import numpy as np
import pytest

from _util import extremes_np, fullrange_np, residual_np, splitmix64

pytestmark = pytest.mark.gpu

CLASSES = [(0, 4), (0, 8), (0, 16), (0, 32), (1, 4), (1, 8), (1, 16)]
# type 2 = DST-VII along rows + DCT-II vertically, type 3 = the other way round (include/x266hip.h)
MIXED = [(2, 4), (2, 8), (2, 16), (3, 4), (3, 8), (3, 16)]


@pytest.mark.parametrize("ttype,n", CLASSES + MIXED)
def test_contiguous_batches(codec, oracle, ttype, n):
    per = n * n
    x = np.concatenate([residual_np(3001 * per, 50 + n), fullrange_np(2000 * per, 51 + n),
                        extremes_np(500 * per, 52 + n)]).reshape(-1, per)
    assert np.array_equal(codec.transform_fwd(ttype, n, x), oracle.transform_fwd(ttype, n, x))


@pytest.mark.parametrize("ttype,n", [c for c in CLASSES + MIXED if c[1] < 32])
@pytest.mark.parametrize("count", [1, 2, 3, 5, 15, 16, 17, 63, 64, 65, 127, 129])
def test_ragged_counts(codec, oracle, ttype, n, count):
    x = fullrange_np(count * n * n, 900 + count + n).reshape(count, n * n)
    assert np.array_equal(codec.transform_fwd(ttype, n, x), oracle.transform_fwd(ttype, n, x))


@pytest.mark.parametrize("ttype,n", CLASSES + MIXED)
def test_inverse_transforms(codec, oracle, ttype, n):
    per = n * n
    r = residual_np(2003 * per, 70 + n).reshape(-1, per)
    z = oracle.transform_fwd(ttype, n, r)
    x = np.concatenate([z, fullrange_np(1000 * per, 71 + n).reshape(-1, per), extremes_np(300 * per, 72 + n).reshape(-1, per)])
    assert np.array_equal(codec.transform_inv(ttype, n, x), oracle.transform_inv(ttype, n, x))
    for count in (1, 2, 3, 15, 17, 63, 65, 129):
        assert np.array_equal(codec.transform_inv(ttype, n, x[:count]), oracle.transform_inv(ttype, n, x[:count]))
    rt = codec.transform_inv(ttype, n, codec.transform_fwd(ttype, n, r))         # round trip on the device
    assert np.abs(rt.astype(np.int32) - r.astype(np.int32)).max() <= 6


def test_edge_blocks(codec, oracle):
    for ttype, n in CLASSES + MIXED:
        per = n * n
        blocks = [np.zeros(per), np.full(per, 255), np.full(per, -256), np.full(per, 32767), np.full(per, -32768),
                  np.where(np.arange(per) % 2 == 0, 32767, -32768), (np.arange(per) % n) * 7 - (np.arange(per) // n) * 3]
        x = np.stack(blocks).astype(np.int16)
        assert np.array_equal(codec.transform_fwd(ttype, n, x), oracle.transform_fwd(ttype, n, x)), (ttype, n)


def _random_ctu_partition(rng_state, n_ctus):
    """Each 64x64 CTU (4096 residual samples, TU-major layout) is cut into a random mix of
    square transform units; returns per-class offset lists (sample offsets into one flat buffer)."""
    r = splitmix64(rng_state, 0, n_ctus * 64)
    offsets = {c: [] for c in CLASSES}
    pos, k = 0, 0
    for ctu in range(n_ctus):
        base, used = ctu * 4096, 0
        while used < 4096:
            pick = int(r[k % len(r)] % 7); k += 1
            ttype, n = CLASSES[pick]
            if used + n * n > 4096:
                ttype, n = 0, 4                                         # fill the remainder with 4x4 DCT units
            offsets[(ttype, n)].append(base + used)
            used += n * n
    return offsets


def test_mixed_per_ctu_batches(codec, oracle):
    """configs[3]: one residual buffer of CTUs, every TU class transformed by its own call
    over an offset table; the whole coefficient buffer must equal the oracle's."""
    n_ctus = 300
    x = np.concatenate([residual_np(n_ctus * 2048, 77), fullrange_np(n_ctus * 2048, 78)])
    offsets = _random_ctu_partition(1234, n_ctus)
    assert sum(len(v) * c[1] * c[1] for c, v in offsets.items()) == n_ctus * 4096
    want = np.zeros_like(x)
    got = np.zeros_like(x)
    din, dout = codec.alloc(x.nbytes), codec.alloc(x.nbytes)
    din.upload(x)
    dout.upload(got)
    keep = []
    for (ttype, n), offs in offsets.items():
        if not offs:
            continue
        offs = np.array(offs, np.uint32)
        idx = offs[:, None] + np.arange(n * n, dtype=np.uint32)[None, :]
        want[idx] = oracle.transform_fwd(ttype, n, x[idx])
        if n == 32:                                                    # 32x32 units of a CTU are gathered first
            blocks = np.ascontiguousarray(x[idx])
            got32 = codec.transform_fwd(0, 32, blocks)
            assert np.array_equal(got32, want[idx])
            continue
        doff = codec.alloc(offs.nbytes)
        doff.upload(offs)
        keep.append(doff)
        codec.transform_fwd_dev(ttype, n, din.ptr, dout.ptr, len(offs), doff.ptr)
    codec.stream_sync()
    got = dout.download(np.int16, x.size)
    for (ttype, n), offs in offsets.items():
        if n == 32 or not offs:
            continue
        idx = np.array(offs, np.uint32)[:, None] + np.arange(n * n, dtype=np.uint32)[None, :]
        assert np.array_equal(got[idx], want[idx]), (ttype, n)


def test_mixed_per_ctu_batches_inverse(codec, oracle):
    """The inverse over the same kind of offset tables (all seven classes, 32x32 included)."""
    n_ctus = 200
    z = np.concatenate([residual_np(n_ctus * 2048, 177), fullrange_np(n_ctus * 2048, 178)])
    offsets = _random_ctu_partition(4321, n_ctus)
    want = np.zeros_like(z)
    din, dout = codec.alloc(z.nbytes), codec.alloc(z.nbytes)
    din.upload(z)
    dout.upload(np.zeros_like(z))
    keep = []
    for (ttype, n), offs in offsets.items():
        if not offs:
            continue
        offs = np.array(offs, np.uint32)
        idx = offs[:, None] + np.arange(n * n, dtype=np.uint32)[None, :]
        want[idx] = oracle.transform_inv(ttype, n, z[idx]) if n < 32 else oracle.dct32_inv(z[idx])
        doff = codec.alloc(offs.nbytes)
        doff.upload(offs)
        keep.append(doff)
        codec.transform_inv_dev(ttype, n, din.ptr, dout.ptr, len(offs), doff.ptr)
    codec.stream_sync()
    assert np.array_equal(dout.download(np.int16, z.size), want)


def test_mixed_per_ctu_forward_32_by_offsets(codec, oracle):
    """(DCT-II, 32) through the offset table directly (no gathering on the host)."""
    x = residual_np(64 * 1024, 9).reshape(64, 1024)
    order = np.random.default_rng(3).permutation(64).astype(np.uint32)
    din, dout, doff = codec.alloc(x.nbytes), codec.alloc(x.nbytes), codec.alloc(order.nbytes)
    din.upload(x)
    doff.upload(order * 1024)
    codec.transform_fwd_dev(0, 32, din.ptr, dout.ptr, 64, doff.ptr)
    codec.stream_sync()
    assert np.array_equal(dout.download(np.int16, x.size).reshape(64, 1024), oracle.dct32_fwd(x))


@pytest.mark.parametrize("n_tiles,with_offsets", [(1, False), (7, False), (403, False), (403, True), (4096, False), (20011, False)])
def test_mixed_tiles_one_launch(codec, oracle, n_tiles, with_offsets):
    """xTransformTilesDev: every tile its own (type, size) class, one launch forward and one inverse, equal to the
    oracle's per-class transforms of the tile's blocks."""
    rng = np.random.default_rng(n_tiles + 17 * with_offsets)
    cls_list = CLASSES + MIXED
    pick = rng.integers(0, len(cls_list), n_tiles)
    tile_class = np.array([t * 4 + {4: 0, 8: 1, 16: 2, 32: 3}[n] for t, n in (cls_list[p] for p in pick)], np.uint8)
    n_slots = n_tiles + (5 if with_offsets else 0)
    x = np.concatenate([residual_np(n_slots * 512, 3), fullrange_np(n_slots * 512, 4)]).astype(np.int16)
    order = rng.permutation(n_slots)[:n_tiles].astype(np.uint32) if with_offsets else np.arange(n_tiles, dtype=np.uint32)
    offsets = order * 1024
    want_f = x.copy() if with_offsets else np.zeros_like(x)
    want_i = want_f.copy()
    for t in range(n_tiles):
        ttype, n = cls_list[pick[t]]
        blk = x[offsets[t]:offsets[t] + 1024].reshape(-1, n * n)
        f = oracle.dct32_fwd(blk) if n == 32 else oracle.transform_fwd(ttype, n, blk)
        want_f[offsets[t]:offsets[t] + 1024] = f.ravel()
        inv = oracle.dct32_inv(f) if n == 32 else oracle.transform_inv(ttype, n, f)
        want_i[offsets[t]:offsets[t] + 1024] = inv.ravel()
    din, dco, dre = codec.alloc(x.nbytes), codec.alloc(x.nbytes), codec.alloc(x.nbytes)
    din.upload(x)
    dco.upload(x if with_offsets else np.zeros_like(x))
    dre.upload(want_f if with_offsets else np.zeros_like(x))
    dcls, doff = codec.alloc(max(n_tiles, 16)), codec.alloc(n_tiles * 4)
    dcls.upload(tile_class)
    doff.upload(offsets)
    codec.transform_tiles_dev(False, din.ptr, dco.ptr, n_tiles, doff.ptr if with_offsets else 0, dcls.ptr)
    codec.transform_tiles_dev(True, dco.ptr, dre.ptr, n_tiles, doff.ptr if with_offsets else 0, dcls.ptr)
    codec.stream_sync()
    assert np.array_equal(dco.download(np.int16, x.size)[: n_tiles * 1024 if not with_offsets else x.size], want_f[: n_tiles * 1024 if not with_offsets else x.size])
    assert np.array_equal(dre.download(np.int16, x.size)[: n_tiles * 1024 if not with_offsets else x.size], want_i[: n_tiles * 1024 if not with_offsets else x.size])


def test_argument_errors(codec):
    L = codec.L
    buf = codec.alloc(1 << 16)
    assert L.xTransformFwdBatchDev(codec.ctx, 0, 5, buf.ptr, buf.ptr + 4096, 4, None, None) < 0
    assert L.xTransformFwdBatchDev(codec.ctx, 1, 32, buf.ptr, buf.ptr + 4096, 1, None, None) < 0      # no DST-VII 32
    assert L.xTransformFwdBatchDev(codec.ctx, 4, 8, buf.ptr, buf.ptr + 4096, 4, None, None) < 0      # types are 0..3
    assert L.xTransformFwdBatchDev(codec.ctx, 2, 32, buf.ptr, buf.ptr + 4096, 1, None, None) < 0     # size 32 is DCT-II only
    assert L.xTransformFwdBatchDev(codec.ctx, 0, 8, None, buf.ptr, 4, None, None) < 0
    assert L.xTransformFwdBatchDev(codec.ctx, 0, 8, None, None, 0, None, None) == 0


def test_transform_set_beyond_4_gib(codec, oracle):
    """Maximum sizes: 2^21 + 3 regions of 1024 samples (4 GiB + 6 KiB per buffer) through every (type, size) class of the set, both directions, and through
    the one-launch tile kernel with the classes cycling: samples at the start, across the 2^32-byte boundary and at the end are bit-exact with the oracle."""
    n_tiles = (1 << 21) + 3
    din, dco, dre = codec.alloc(n_tiles * 2048), codec.alloc(n_tiles * 2048), codec.alloc(n_tiles * 2048)
    codec.fill_residual_dev(din.ptr, n_tiles * 1024, 0x7E5)

    def tiles(buf, first, count):
        out = np.empty(count * 1024, np.int16)
        codec._check(codec.L.xHipMemcpyD2H(codec.ctx, out.ctypes.data, buf.ptr + first * 2048, out.nbytes), "D2H")
        return out.reshape(count, 1024)

    spots = [(0, 6), ((1 << 21) - 3, 6), (n_tiles - 5, 5)]                       # tile 2^21 starts at byte 2^32
    for ttype, n in [(0, 4), (0, 8), (0, 16), (1, 4), (1, 8), (1, 16)]:
        nb = n_tiles * (1024 // (n * n))
        codec.transform_fwd_dev(ttype, n, din.ptr, dco.ptr, nb)
        codec.transform_inv_dev(ttype, n, dco.ptr, dre.ptr, nb)
        codec.stream_sync()
        for first, count in spots:
            f = oracle.transform_fwd(ttype, n, tiles(din, first, count).reshape(-1, n * n))
            assert np.array_equal(tiles(dco, first, count).reshape(-1, n * n), f), (ttype, n, first)
            assert np.array_equal(tiles(dre, first, count).reshape(-1, n * n), oracle.transform_inv(ttype, n, f)), (ttype, n, first)
    q = np.arange(n_tiles)
    pick = ((q * 5 + q // 7) % len(CLASSES)).astype(np.int64)
    tile_class = np.array([t * 4 + {4: 0, 8: 1, 16: 2, 32: 3}[n] for t, n in CLASSES], np.uint8)[pick]
    dcls = codec.alloc(n_tiles)
    dcls.upload(tile_class)
    codec.transform_tiles_dev(False, din.ptr, dco.ptr, n_tiles, 0, dcls.ptr)
    codec.transform_tiles_dev(True, dco.ptr, dre.ptr, n_tiles, 0, dcls.ptr)
    codec.stream_sync()
    for first, count in spots:
        x, got_f, got_i = tiles(din, first, count), tiles(dco, first, count), tiles(dre, first, count)
        for k in range(count):
            ttype, n = CLASSES[pick[first + k]]
            blk = x[k].reshape(-1, n * n)
            f = oracle.dct32_fwd(blk) if n == 32 else oracle.transform_fwd(ttype, n, blk)
            inv = oracle.dct32_inv(f) if n == 32 else oracle.transform_inv(ttype, n, f)
            assert np.array_equal(got_f[k], f.ravel()) and np.array_equal(got_i[k], inv.ravel()), (first + k, ttype, n)
