"""GPU (-m gpu): caller-supplied 1-D transform matrices (xTransformSetMatrix).  The RTL re-uses one datapath for any
tap set (src/mkDct32.bsv:132-141, 385-387); here an integrator who holds other integer tables (the normative H.266
DST-VII for N = 8 / 16, DCT-VIII, ...) installs them per context.  Checked against the oracle's passes with the same
matrices through every entry point of the set: per-class forward and inverse calls and the one-launch tile call."""
import numpy as np
import pytest

import x266_amd
from _util import extremes_np, fullrange_np, residual_np

pytestmark = pytest.mark.gpu

# H.266's DST-VII integers for N = 8 and 16 AS RECALLED from the VTM sources (DEFINE_DST7_P8_MATRIX /
# DEFINE_DST7_P16_MATRIX) -- UNVERIFIED: there is no copy of the standard in this environment.  They have the
# closed form's sign / index pattern with hand-tuned magnitudes, so they are derived from the built-in tables by
# magnitude substitution.  Whatever their standing, they are "a different matrix" for the purpose of these tests.
RECALLED = {8: ((16, 32, 46, 59, 70, 79, 84, 87), (17, 32, 46, 60, 71, 78, 85, 86)),
            16: ((8, 17, 25, 33, 41, 48, 55, 62, 67, 73, 77, 81, 84, 87, 88, 89), (8, 17, 25, 33, 40, 48, 55, 62, 68, 73, 77, 81, 85, 87, 88, 88))}


@pytest.fixture()
def fresh():
    """A context of its own: installed matrices are per context and must not leak into the other tests' codec."""
    c = x266_amd.Codec(0)
    yield c
    c.close()


def recalled_dst7(codec, n):
    base = codec.get_transform_matrix(1, n).astype(np.int16)
    closed, tuned = RECALLED[n]
    out = np.zeros_like(base)
    for a, b in zip(closed, tuned):
        out[np.abs(base) == a] = b
    return (out * np.sign(base)).astype(np.int8)


def check_everywhere(codec, oracle, n, m0, m1):
    """m0 / m1: the int16 matrices now in slot 0 / 1 at size n."""
    per = n * n
    x = np.concatenate([residual_np(1501 * per, 60 + n), fullrange_np(700 * per, 61 + n), extremes_np(200 * per, 62 + n)]).reshape(-1, per)
    slots = (m0, m1)
    for ttype, (hs, vs) in enumerate(((0, 0), (1, 1), (1, 0), (0, 1))):          # type -> (horizontal slot, vertical slot)
        f = oracle.transform_matrix_passes(slots[hs], slots[vs], x)
        assert np.array_equal(codec.transform_fwd(ttype, n, x), f), ("fwd", ttype, n)
        assert np.array_equal(codec.transform_inv(ttype, n, f), oracle.transform_matrix_passes(slots[hs], slots[vs], f, inverse=True)), ("inv", ttype, n)
    # the one-launch tile call: tiles of all four types at this size, next to untouched classes
    n_tiles = 257
    rng = np.random.default_rng(n)
    types = rng.integers(0, 4, n_tiles)
    l = {4: 0, 8: 1, 16: 2}[n]
    tile_class = (types * 4 + l).astype(np.uint8)
    tile_class[::5] = 3                                                           # DCT-II 32 tiles in between
    xt = np.concatenate([residual_np(n_tiles * 512, 7), fullrange_np(n_tiles * 512, 8)]).astype(np.int16)
    want_f, want_i = np.empty_like(xt), np.empty_like(xt)
    for t in range(n_tiles):
        blk = xt[t * 1024:(t + 1) * 1024]
        if tile_class[t] == 3:
            f = oracle.dct32_fwd(blk.reshape(1, 1024)); i = oracle.dct32_inv(f)
        else:
            hs, vs = ((0, 0), (1, 1), (1, 0), (0, 1))[types[t]]
            f = oracle.transform_matrix_passes(slots[hs], slots[vs], blk.reshape(-1, per))
            i = oracle.transform_matrix_passes(slots[hs], slots[vs], f, inverse=True)
        want_f[t * 1024:(t + 1) * 1024] = f.ravel(); want_i[t * 1024:(t + 1) * 1024] = i.ravel()
    din, dco, dre, dcls = codec.alloc(xt.nbytes), codec.alloc(xt.nbytes), codec.alloc(xt.nbytes), codec.alloc(max(n_tiles, 16))
    din.upload(xt); dcls.upload(tile_class)
    codec.transform_tiles_dev(False, din.ptr, dco.ptr, n_tiles, 0, dcls.ptr)
    codec.transform_tiles_dev(True, dco.ptr, dre.ptr, n_tiles, 0, dcls.ptr)
    codec.stream_sync()
    assert np.array_equal(dco.download(np.int16, xt.size), want_f) and np.array_equal(dre.download(np.int16, xt.size), want_i)


@pytest.mark.parametrize("n", [4, 8, 16])
@pytest.mark.parametrize("slot", [0, 1])
def test_random_int8_matrix(fresh, oracle, n, slot):
    """Any int8 matrix, including -128 and 127 entries: the kernels only need the operand images rebuilt."""
    rng = np.random.default_rng(100 * n + slot)
    m = rng.integers(-128, 128, (n, n)).astype(np.int8)
    m[0, 0], m[n - 1, n - 1] = -128, 127
    other = fresh.get_transform_matrix(1 - slot, n).astype(np.int16)
    fresh.set_transform_matrix(slot, n, m)
    assert np.array_equal(fresh.get_transform_matrix(slot, n), m)
    mats = (m.astype(np.int16), other) if slot == 0 else (other, m.astype(np.int16))
    check_everywhere(fresh, oracle, n, *mats)


@pytest.mark.parametrize("n", [8, 16])
def test_recalled_h266_dst7(fresh, oracle, n):
    m = recalled_dst7(fresh, n)
    assert (m != fresh.get_transform_matrix(1, n)).any()                          # it IS a different matrix
    assert abs(int(m.astype(np.int32)[0] @ m.astype(np.int32)[1])) < 64 * 64      # and still a near-orthogonal basis
    fresh.set_transform_matrix(1, n, m)
    check_everywhere(fresh, oracle, n, fresh.get_transform_matrix(0, n).astype(np.int16), m.astype(np.int16))


@pytest.mark.parametrize("n", [4, 8, 16])
def test_named_presets(fresh, oracle, n):
    """xTransformUsePreset: the recalled VTM DST-VII and the DCT-VIII derived from it (T8[k][c] = (-1)^k T7[k][N-1-c]) in slot 1 at
    all sizes at once; every entry point of the set then equals the oracle's passes with the same matrices; preset 0 restores."""
    assert fresh.transform_preset() == 0
    closed = fresh.get_transform_matrix(1, n).copy()
    dct2 = fresh.get_transform_matrix(0, n).astype(np.int16)
    t7 = recalled_dst7(fresh, n) if n > 4 else closed                              # N = 4: the closed form is the standard's table
    fresh.use_transform_preset(1)
    assert fresh.transform_preset() == 1 and np.array_equal(fresh.get_transform_matrix(1, n), t7)
    assert np.array_equal(fresh.get_transform_matrix(0, n).astype(np.int16), dct2)   # slot 0 untouched
    check_everywhere(fresh, oracle, n, dct2, t7.astype(np.int16))
    fresh.use_transform_preset(2)
    t8 = (t7[:, ::-1].astype(np.int16) * ((-1) ** np.arange(n))[:, None]).astype(np.int8)
    assert fresh.transform_preset() == 2 and np.array_equal(fresh.get_transform_matrix(1, n), t8)
    g = t8.astype(np.int32) @ t8.astype(np.int32).T                                  # a near-orthogonal basis of norm ~ 64 sqrt(N) like the others
    assert abs(g - np.diag(np.diag(g))).max() < 0.05 * g.diagonal().min()
    check_everywhere(fresh, oracle, n, dct2, t8.astype(np.int16))
    fresh.set_transform_matrix(1, n, t7)
    assert fresh.transform_preset() == -1                                            # a caller's own matrix is not a preset
    fresh.use_transform_preset(0)
    assert fresh.transform_preset() == 0 and np.array_equal(fresh.get_transform_matrix(1, n), closed)
    with pytest.raises(x266_amd.X266Error):
        fresh.use_transform_preset(3)


def test_defaults_and_restore(fresh, oracle, codec):
    """Built-ins = the oracle's tables; NULL restores them; other contexts are not affected; bad arguments are rejected."""
    for n in (4, 8, 16):
        assert np.array_equal(fresh.get_transform_matrix(0, n), oracle.transform_matrix(0, n))
        assert np.array_equal(fresh.get_transform_matrix(1, n), oracle.transform_matrix(1, n))
    n = 8
    x = residual_np(777 * 64, 5).reshape(-1, 64)
    before = fresh.transform_fwd(1, n, x)
    assert np.array_equal(before, oracle.transform_fwd(1, n, x))
    fresh.set_transform_matrix(1, n, np.eye(n, dtype=np.int8) * 64)
    assert not np.array_equal(fresh.transform_fwd(1, n, x), before)
    assert np.array_equal(codec.transform_fwd(1, n, x), before)                   # the session's shared context still has the built-in
    fresh.set_transform_matrix(1, n, None)
    assert np.array_equal(fresh.transform_fwd(1, n, x), before)
    assert np.array_equal(fresh.transform_fwd(0, 32, residual_np(64 * 1024, 6).reshape(-1, 1024)), oracle.dct32_fwd(residual_np(64 * 1024, 6).reshape(-1, 1024)))
    L = fresh.L
    assert L.xTransformSetMatrix(fresh.ctx, 0, 32, None) < 0                       # the pinned 32-point DCT-II cannot be replaced
    assert L.xTransformSetMatrix(fresh.ctx, 2, 8, None) < 0 and L.xTransformSetMatrix(fresh.ctx, 1, 5, None) < 0
