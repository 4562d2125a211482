"""GPU: xHipMemCeilingDev -- the arithmetic-free streams bench.py and the perf-floor tests normalise by.  They are diagnostics, but
their numbers only mean something if the streams really move every byte: the copy is compared, the read stream's per-2 KiB XOR
checksums are recomputed on the host, the write stream's pattern is checked; ragged sizes included."""
import numpy as np
import pytest

import x266_amd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nbytes", [16, 2048, 2048 + 16, 4096, 4096 * 3 + 1024 + 48, 1 << 20, (1 << 22) + 2048 * 3 + 16])
def test_streams_move_every_byte(codec, nbytes):
    rs = np.random.RandomState(nbytes % 9973)
    src = rs.randint(0, 1 << 32, size=nbytes // 4, dtype=np.uint64).astype(np.uint32)
    pieces = (nbytes + 2047) // 2048
    d_src, d_dst, d_sum = codec.alloc(nbytes), codec.alloc(nbytes + 16), codec.alloc(pieces * 4 + 16)
    d_src.upload(src)
    d_dst.upload(np.zeros(nbytes // 4 + 4, np.uint32))
    codec.mem_ceiling_dev(0, d_src.ptr, d_dst.ptr, nbytes)
    codec.mem_ceiling_dev(1, d_src.ptr, d_sum.ptr, nbytes)
    codec.stream_sync()
    got = d_dst.download(np.uint32, nbytes // 4 + 4)
    assert np.array_equal(got[:-4], src) and not got[-4:].any()                 # nothing written past the end
    want = np.zeros(pieces, np.uint32)
    padded = np.zeros(pieces * 512, np.uint32)
    padded[: src.size] = src
    want = np.bitwise_xor.reduce(padded.reshape(pieces, 512), axis=1)
    assert np.array_equal(d_sum.download(np.uint32, pieces), want)
    # the probe reads the same bytes and stores nothing -- unless a piece's XOR is the magic: plant it in the last whole piece
    d_sum.upload(np.zeros(pieces + 4, np.uint32))
    codec.mem_ceiling_dev(3, d_src.ptr, d_sum.ptr, nbytes)
    codec.stream_sync()
    assert not d_sum.download(np.uint32, pieces).any()
    if nbytes >= 2048:
        hit = nbytes // 2048 - 1
        src2 = src.copy()
        src2[hit * 512] ^= want[hit] ^ np.uint32(0x12345678)
        d_src.upload(src2)
        codec.mem_ceiling_dev(3, d_src.ptr, d_sum.ptr, nbytes)
        codec.stream_sync()
        got3 = d_sum.download(np.uint32, pieces)
        assert got3[hit] == 0x12345678 and not np.delete(got3, hit).any()
    codec.mem_ceiling_dev(2, 0, d_dst.ptr, nbytes)
    codec.stream_sync()
    got = d_dst.download(np.uint32, nbytes // 4 + 4)
    pat = np.zeros(nbytes // 4, np.uint32)
    pat[0::4] = np.arange(nbytes // 16, dtype=np.uint32)
    assert np.array_equal(got[:-4], pat) and not got[-4:].any()


def test_bad_arguments_are_refused(codec):
    d = codec.alloc(4096)
    for kind, src, dst, n in ((4, d.ptr, d.ptr, 64), (0, d.ptr + 4, d.ptr + 2048, 64), (0, d.ptr, d.ptr + 2048, 24), (1, 0, d.ptr, 64)):
        with pytest.raises(x266_amd.X266Error):
            codec.mem_ceiling_dev(kind, src, dst, n)
