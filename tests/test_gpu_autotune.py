"""GPU (-m gpu): the opt-in "autotune" option (include/x266hip.h).  Results never depend on it: every family is run with the option
off and on, on the same inputs, and must write the same bytes -- whatever candidate shape the box made it keep; small batches, knobs set
by the caller and overlapping buffers leave it untuned."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(codec, a, b, nbytes, chunk=256 << 20):
    for off in range(0, nbytes, chunk):
        n = min(chunk, nbytes - off)
        ha, hb = np.empty(n, np.uint8), np.empty(n, np.uint8)
        codec._check(codec.L.xHipMemcpyD2H(codec.ctx, ha.ctypes.data, a.ptr + off, n), "D2H")
        codec._check(codec.L.xHipMemcpyD2H(codec.ctx, hb.ctypes.data, b.ptr + off, n), "D2H")
        if not np.array_equal(ha, hb):
            return False
    return True


@pytest.fixture
def tuned_codec():
    import x266_amd
    c = x266_amd.Codec(0)                                                # its own context: tuned state must not leak into the shared fixture
    yield c
    c.close()


def test_fused_forward_inverse_same_bytes_with_any_chosen_shape(tuned_codec, oracle):
    codec = tuned_codec
    n = (1 << 18) + 5                                                    # large enough to tune, ragged
    x, z0, r0, z1, r1 = (codec.alloc(n * 2048) for _ in range(5))
    codec.fill_residual_dev(x.ptr, n * 1024, 0x51)
    codec.dct32_fwd_inv_dev(x.ptr, z0.ptr, r0.ptr, n)
    codec.stream_sync()
    assert codec.autotune_report() == {}
    codec.set_option("autotune", 1)
    codec.dct32_fwd_inv_dev(x.ptr, z1.ptr, r1.ptr, 4096)                 # small: not tuned by this call
    assert "dct32_fwd_inv" not in codec.autotune_report()
    codec.dct32_fwd_inv_dev(x.ptr, z1.ptr, r1.ptr, n)
    codec.stream_sync()
    rep = codec.autotune_report()["dct32_fwd_inv"]
    assert len(rep["ms"]) == 8 and all(m > 0 for m in rep["ms"]) and 0 <= rep["choice"] < 8
    assert rep["choice"] == 0 or rep["ms"][rep["choice"]] < 0.98 * rep["ms"][0]      # the default stays unless beaten by more than 2 %
    assert _same(codec, z0, z1, n * 2048) and _same(codec, r0, r1, n * 2048)
    head = z1.download(np.int16, 64 * 1024).reshape(64, 1024)
    assert np.array_equal(head, oracle.dct32_fwd(oracle.fill_residual(64 * 1024, 0x51)))
    # every candidate, forced in turn through a fresh tuned state, writes the same bytes (the shapes differ in pipeline depth and run length)
    codec.dct32_fwd_inv_dev(x.ptr, 0, r1.ptr, n)                         # reconstruction only: its own family
    codec.stream_sync()
    assert "dct32_recon_only" in codec.autotune_report() and _same(codec, r0, r1, n * 2048)
    # the caller's own knob wins over the tuner; overlapping buffers are never tuned on
    codec.set_option("dct32_fwdinv_blocks_per_wave", 3)
    codec.dct32_fwd_inv_dev(x.ptr, z1.ptr, r1.ptr, n)
    codec.stream_sync()
    assert _same(codec, z0, z1, n * 2048) and _same(codec, r0, r1, n * 2048)


def test_satd_and_sad_batches_same_bytes(tuned_codec):
    codec = tuned_codec
    n = (1 << 23) + 77
    d, s0, s1 = codec.alloc(n * 128), codec.alloc(n * 4), codec.alloc(n * 4)
    codec.fill_residual_dev(d.ptr, n * 64, 0x52)
    codec.satd8x8_dev(d.ptr, s0.ptr, n)
    codec.set_option("autotune", 1)
    codec.satd8x8_dev(d.ptr, s1.ptr, n)
    codec.stream_sync()
    rep = codec.autotune_report()
    assert "satd8x8" in rep and all(m > 0 for m in rep["satd8x8"]["ms"])
    assert _same(codec, s0, s1, n * 4)
    for edge in (8, 16, 32, 64):                                         # the residual bytes seen as 8-bit blocks
        nb = n * 128 // 2 // (edge * edge)
        codec.set_option("autotune", 0)
        codec.sad_dev(edge, d.ptr, d.ptr + n * 64, s0.ptr, nb)
        codec.set_option("autotune", 1)
        codec.sad_dev(edge, d.ptr, d.ptr + n * 64, s1.ptr, nb)
        codec.stream_sync()
        assert "sad%d" % edge in codec.autotune_report()
        assert _same(codec, s0, s1, nb * 4), edge


def test_not_under_stream_capture(tuned_codec):
    codec = tuned_codec
    n = 1 << 18
    x, z, r = codec.alloc(n * 2048), codec.alloc(n * 2048), codec.alloc(n * 2048)
    codec.fill_residual_dev(x.ptr, n * 1024, 0x53)
    codec.set_option("autotune", 1)
    st = codec.stream_create()
    codec.stream_sync()
    codec.graph_begin(st)
    codec.dct32_fwd_inv_dev(x.ptr, z.ptr, r.ptr, n, st)                  # a capture cannot be timed: the default shape is recorded
    g = codec.graph_end(st)
    assert codec.autotune_report() == {}
    codec.graph_launch(g, st)
    codec.stream_sync(st)
    codec.graph_free(g)
    codec.stream_destroy(st)
