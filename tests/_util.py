"""Shared helpers for the test-suite: ctypes loaders for the oracle (checker),
the optional real-reference build (oracle/_ref) and small numpy utilities.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch
anything under oracle/ -- it is the checker, never the product.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

_P = ctypes.c_void_p
_SZ = ctypes.c_size_t


def _build_oracle():
    so = os.path.join(ORACLE_DIR, "liborc.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "--no-print-directory", so])
    return so


class Oracle:
    """ctypes view of oracle/liborc.so (this repo's C restatement)."""

    def __init__(self):
        self.lib = ctypes.CDLL(_build_oracle())
        L = self.lib
        L.orc_dct32_table.restype = ctypes.POINTER(ctypes.c_int16)
        L.orc_satd8x8.restype = ctypes.c_uint32
        L.orc_pack_dct_word.restype = ctypes.c_uint64
        L.orc_checksum64.restype = ctypes.c_uint64
        L.orc_sum_u16.restype = ctypes.c_uint64
        L.orc_hw_threads.restype = ctypes.c_int

    def table(self):
        return np.ctypeslib.as_array(self.lib.orc_dct32_table(), (32, 32)).copy()

    def dct32_fwd(self, x, threads=1):
        x = np.ascontiguousarray(x, np.int16).reshape(-1, 1024)
        out = np.empty_like(x)
        self.lib.orc_dct32_fwd_mt(_P(x.ctypes.data), _P(out.ctypes.data), _SZ(x.shape[0]), threads)
        return out

    def dct32_inv(self, z, threads=1):
        z = np.ascontiguousarray(z, np.int16).reshape(-1, 1024)
        out = np.empty_like(z)
        self.lib.orc_dct32_inv_mt(_P(z.ctypes.data), _P(out.ctypes.data), _SZ(z.shape[0]), threads)
        return out

    def dct32_pass(self, src, shift, line=32, dense=False):
        src = np.ascontiguousarray(src, np.int16)
        dst = np.empty(32 * line, np.int16)
        fn = self.lib.orc_dct32_pass_dense if dense else self.lib.orc_dct32_pass
        fn(_P(src.ctypes.data), _P(dst.ctypes.data), shift, line)
        return dst

    def satd8x8(self, d, threads=1):
        d = np.ascontiguousarray(d, np.int16).reshape(-1, 64)
        out = np.empty(d.shape[0], np.uint32)
        self.lib.orc_satd8x8_batch_mt(_P(d.ctypes.data), _P(out.ctypes.data), _SZ(d.shape[0]), threads)
        return out

    def conv_input_fmt(self, y, u, v):
        y = np.ascontiguousarray(y, np.uint8)
        u = np.ascontiguousarray(u, np.uint8)
        v = np.ascontiguousarray(v, np.uint8)
        h, w = y.shape
        tiles = np.zeros((h // 16) * (w // 16) * 512, np.uint8)
        assert u.strides[0] == y.strides[0] // 2 and v.strides[0] == u.strides[0]
        self.lib.orc_conv_input_fmt(_P(tiles.ctypes.data), _P(y.ctypes.data), _P(u.ctypes.data), _P(v.ctypes.data),
                                    ctypes.c_ssize_t(y.strides[0]), w, h)
        return tiles

    def conv_output_420(self, tiles, w, h):
        tiles = np.ascontiguousarray(tiles, np.uint8)
        y = np.zeros((h, w), np.uint8)
        u = np.zeros((h // 2, w // 2), np.uint8)
        v = np.zeros((h // 2, w // 2), np.uint8)
        self.lib.orc_conv_output_420(_P(tiles.ctypes.data), _P(y.ctypes.data), ctypes.c_ssize_t(w), _P(u.ctypes.data),
                                     _P(v.ctypes.data), ctypes.c_ssize_t(w // 2), w, h)
        return y, u, v

    def residual_luma(self, cur_tiles, pred_tiles, w, h, edge):
        cur_tiles = np.ascontiguousarray(cur_tiles, np.uint8)
        pred_tiles = np.ascontiguousarray(pred_tiles, np.uint8)
        res = np.empty(w * h, np.int16)
        self.lib.orc_residual_luma(_P(cur_tiles.ctypes.data), _P(pred_tiles.ctypes.data), w, h, edge, _P(res.ctypes.data))
        return res

    def residual_chroma(self, cur_tiles, pred_tiles, w, h, edge, block_pitch=1):
        """(res_u, res_v): block_pitch 1 -> two planar block streams; block_pitch 2 -> views into one CTU-ordered buffer U0 V0 U1 V1 ..."""
        cur_tiles = np.ascontiguousarray(cur_tiles, np.uint8)
        pred_tiles = np.ascontiguousarray(pred_tiles, np.uint8)
        n = (w // 2) * (h // 2)
        if block_pitch == 1:
            res_u, res_v = np.empty(n, np.int16), np.empty(n, np.int16)
            pu, pv = res_u.ctypes.data, res_v.ctypes.data
        else:
            both = np.full(n * block_pitch, 0x7777, np.int16)
            res_u, res_v = both, both[edge * edge:]
            pu, pv = both.ctypes.data, both.ctypes.data + edge * edge * 2
        self.lib.orc_residual_chroma(_P(cur_tiles.ctypes.data), _P(pred_tiles.ctypes.data), w, h, edge, _P(pu), _P(pv), _SZ(block_pitch))
        return res_u, res_v

    def transform_matrix(self, ttype, n):
        m = np.empty((n, n), np.int16)
        assert self.lib.orc_transform_matrix(ttype, n, _P(m.ctypes.data)) == 0
        return m

    def transform_fwd(self, ttype, n, x):
        x = np.ascontiguousarray(x, np.int16).reshape(-1, n * n)
        out = np.empty_like(x)
        assert self.lib.orc_transform_fwd(ttype, n, _P(x.ctypes.data), _P(out.ctypes.data), _SZ(x.shape[0])) == 0
        return out

    def transform_inv(self, ttype, n, x):
        x = np.ascontiguousarray(x, np.int16).reshape(-1, n * n)
        out = np.empty_like(x)
        assert self.lib.orc_transform_inv(ttype, n, _P(x.ctypes.data), _P(out.ctypes.data), _SZ(x.shape[0])) == 0
        return out

    def transform_matrix_passes(self, mh, mv, x, inverse=False):
        """The set's two passes with caller-supplied n x n matrices (mh along rows, mv vertically; row k = basis function)."""
        mh = np.ascontiguousarray(mh, np.int16)
        mv = np.ascontiguousarray(mv, np.int16)
        n = mh.shape[0]
        x = np.ascontiguousarray(x, np.int16).reshape(-1, n * n)
        out = np.empty_like(x)
        fn = self.lib.orc_transform_inv_matrix if inverse else self.lib.orc_transform_fwd_matrix
        assert fn(_P(mh.ctypes.data), _P(mv.ctypes.data), n, _P(x.ctypes.data), _P(out.ctypes.data), _SZ(x.shape[0])) == 0
        return out

    def intra32_predict(self, refs, modes, ref_index=None):
        """refs [n_refs,129] uint8 (left[64] | top[65]); modes [n]; ref_index [n] or None -> [n,1024] uint8."""
        refs = np.ascontiguousarray(refs, np.uint8).reshape(-1, 129)
        modes = np.ascontiguousarray(modes, np.uint8)
        ri = None if ref_index is None else np.ascontiguousarray(ref_index, np.uint32)
        out = np.empty((modes.shape[0], 1024), np.uint8)
        rc = self.lib.orc_intra32_predict_batch(_P(refs.ctypes.data), _P(modes.ctypes.data), _P(ri.ctypes.data if ri is not None else None),
                                                _P(out.ctypes.data), _SZ(modes.shape[0]))
        assert rc == 0
        return out

    def intra32_costs(self, refs, src):
        """refs [n,129], src [n,1024] uint8 -> (costs [n,35] uint32, best_mode [n] uint8)."""
        refs = np.ascontiguousarray(refs, np.uint8).reshape(-1, 129)
        src = np.ascontiguousarray(src, np.uint8).reshape(-1, 1024)
        n = refs.shape[0]
        costs = np.empty((n, 35), np.uint32)
        best = np.empty(n, np.uint8)
        assert self.lib.orc_intra32_costs(_P(refs.ctypes.data), _P(src.ctypes.data), _SZ(n), _P(costs.ctypes.data), _P(best.ctypes.data)) == 0
        return costs, best

    def satd_search(self, cur, ref_padded, pad, rng, threads=1, want_costs=False, metric="satd"):
        """cur [H,W] uint8; ref_padded [H+2*pad, W+2*pad] uint8 with pad >= rng.  metric: "satd" or "sad"."""
        cur = np.ascontiguousarray(cur, np.uint8)
        refp = np.ascontiguousarray(ref_padded, np.uint8)
        h, w = cur.shape
        nb = (h // 8) * (w // 8)
        mv = np.empty((nb, 2), np.int16)
        cost = np.empty(nb, np.uint32)
        costs = np.empty((nb, (2 * rng + 1) ** 2), np.uint32) if want_costs else None
        origin = refp.ctypes.data + pad * refp.strides[0] + pad
        fn = self.lib.orc_satd8x8_search if metric == "satd" else self.lib.orc_sad8x8_search
        fn(_P(cur.ctypes.data), ctypes.c_ssize_t(cur.strides[0]), _P(origin),
                                    ctypes.c_ssize_t(refp.strides[0]), w, h, rng, _P(mv.ctypes.data),
                                    _P(cost.ctypes.data), _P(costs.ctypes.data if want_costs else None), threads)
        return mv, cost, costs

    def fill_residual(self, n_samples, seed, first_index=0):
        out = np.empty(n_samples, np.int16)
        self.lib.orc_fill_residual(_P(out.ctypes.data), _SZ(n_samples), ctypes.c_uint64(seed),
                                   ctypes.c_uint64(first_index))
        return out

    def pack_diff_rows(self, mat, first_row):
        mat = np.ascontiguousarray(mat, np.int16)
        res = np.empty(32, np.uint32)
        self.lib.orc_pack_diff_rows(_P(mat.ctypes.data), first_row, _P(res.ctypes.data))
        return res

    def pack_dct_word(self, dct, idx):
        dct = np.ascontiguousarray(dct, np.int16)
        return int(self.lib.orc_pack_dct_word(_P(dct.ctypes.data), idx))

    def checksum64(self, a):
        a = np.ascontiguousarray(a)
        return int(self.lib.orc_checksum64(_P(a.ctypes.data), _SZ(a.nbytes)))

    def hw_threads(self):
        return int(self.lib.orc_hw_threads())


def ref_path():
    return os.path.join(ORACLE_DIR, "_ref", "libx266ref.so")


class Reference:
    """ctypes view of oracle/_ref/libx266ref.so -- the REAL reference golden
    models (src_tb/dct32.c, satd.c) compiled from /root/reference in place.
    Present only where `make -C oracle ref` has run (not in git)."""

    def __init__(self):
        self.lib = ctypes.CDLL(ref_path())
        L = self.lib
        L.ref_dct32_table.restype = ctypes.POINTER(ctypes.c_int16)
        L.ref_dct32_last_input.restype = ctypes.POINTER(ctypes.c_int16)
        L.ref_dct32_last_output.restype = ctypes.POINTER(ctypes.c_int16)
        L.ref_satd8x8_last_input.restype = ctypes.POINTER(ctypes.c_int16)
        L.dct32_getDct.restype = ctypes.c_uint64
        L.satd8x8_getSatd.restype = ctypes.c_uint32

    def table(self):
        return np.ctypeslib.as_array(self.lib.ref_dct32_table(), (32, 32)).copy()

    def dct32_fwd(self, x):
        x = np.ascontiguousarray(x, np.int16).reshape(-1, 1024)
        out = np.empty_like(x)
        self.lib.ref_dct32_fwd(_P(x.ctypes.data), _P(out.ctypes.data), ctypes.c_ulong(x.shape[0]))
        return out

    def dct32_pass(self, src, shift, line=32):
        src = np.ascontiguousarray(src, np.int16)
        dst = np.empty(32 * line, np.int16)
        self.lib.ref_dct32_pass(_P(src.ctypes.data), _P(dst.ctypes.data), shift, line)
        return dst

    def satd8x8(self, d):
        d = np.ascontiguousarray(d, np.int16).reshape(-1, 64)
        out = np.empty(d.shape[0], np.uint32)
        self.lib.ref_satd8x8_batch(_P(d.ctypes.data), _P(out.ctypes.data), ctypes.c_ulong(d.shape[0]))
        return out


# --------------------------------------------------------------------------
# numpy twin of the counter-based SplitMix64 stream (oracle/prng_oracle.c and
# the device kernel x266_fill_residual); used to make full-range test data.
# --------------------------------------------------------------------------
def splitmix64(seed, first_index, count):
    with np.errstate(over="ignore"):
        idx = np.arange(first_index + 1, first_index + 1 + count, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def residual_np(n_samples, seed, first_index=0):
    r = splitmix64(seed, first_index, n_samples)
    return ((r & np.uint64(0xFF)).astype(np.int32) - ((r >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int32)).astype(np.int16)


def fullrange_np(n_samples, seed, first_index=0):
    r = splitmix64(seed, first_index, n_samples)
    return (r >> np.uint64(16)).astype(np.uint16).view(np.int16)


def extremes_np(n_samples, seed, first_index=0):
    r = splitmix64(seed, first_index, n_samples)
    return np.where((r >> np.uint64(40)) & np.uint64(1), np.int16(32767), np.int16(-32768)).astype(np.int16)


def me_frames(w, h, pad, seed, mv=(3, -2), noise=6):
    """Synthetic frame pair: smooth-ish random cur; ref = cur displaced by a known global
    motion vector plus noise, padded by `pad` (edge replication).  uint8."""
    M = 16                                                            # margin: |mv| <= 16
    assert max(abs(mv[0]), abs(mv[1])) <= M
    r = splitmix64(seed, 0, (h + 2 * pad + 2 * M) * (w + 2 * pad + 2 * M))
    base = (r & np.uint64(0xFF)).astype(np.float64).reshape(h + 2 * pad + 2 * M, w + 2 * pad + 2 * M)
    k = np.ones(5) / 5.0                                             # separable low-pass so that motion is findable
    base = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, base)
    base = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, base)
    base = np.clip((base - 128.0) * 3.0 + 128.0, 0, 255)
    big = base.astype(np.uint8)
    cur = big[pad + M:pad + M + h, pad + M:pad + M + w].copy()
    # ref(x + mvx, y + mvy) == cur(x, y)  =>  the best displacement is mv
    refp = big[M - mv[1]:M - mv[1] + h + 2 * pad, M - mv[0]:M - mv[0] + w + 2 * pad].astype(np.int16)
    nz = splitmix64(seed + 1, 0, refp.size)
    refp = refp + ((nz & np.uint64(0xFF)).astype(np.int16).reshape(refp.shape) % (2 * noise + 1) - noise)
    return cur, np.clip(refp, 0, 255).astype(np.uint8)


def dct_edge_blocks():
    """Input-independent corner cases for the 32x32 transform."""
    blocks, names = [], []

    def add(name, b):
        names.append(name)
        blocks.append(np.asarray(b, np.int16).reshape(1024))

    add("zeros", np.zeros(1024))
    add("all_255", np.full(1024, 255))
    add("all_m256", np.full(1024, -256))
    add("all_32767", np.full(1024, 32767))
    add("all_m32768", np.full(1024, -32768))
    alt = np.where(np.arange(1024) % 2 == 0, 32767, -32768)
    add("alt_cols_extreme", alt)
    rows = np.where((np.arange(1024) // 32) % 2 == 0, 32767, -32768)
    add("alt_rows_extreme", rows)
    chk = np.where(((np.arange(1024) // 32) + np.arange(1024)) % 2 == 0, 255, -255)
    add("checker_255", chk)
    for pos in (0, 31, 32 * 31, 1023, 32 * 7 + 19):
        imp = np.zeros(1024)
        imp[pos] = 255
        add("impulse_%d" % pos, imp)
    ramp = (np.arange(1024) % 32) * 16 - 248
    add("h_ramp", ramp)
    vramp = (np.arange(1024) // 32) * 16 - 248
    add("v_ramp", vramp)
    asym = (np.arange(1024) % 32) * 7 - (np.arange(1024) // 32) * 3   # transpose-detecting
    add("asymmetric", asym)
    return np.stack(blocks), names


def satd_edge_blocks():
    blocks, names = [], []

    def add(name, b):
        names.append(name)
        blocks.append(np.asarray(b, np.int16).reshape(64))

    add("zeros", np.zeros(64))                     # -> 0
    add("all_255", np.full(64, 255))               # -> 4080
    add("all_m256", np.full(64, -256))             # -> 4096
    add("all_32767", np.full(64, 32767))           # -> 16 (int16 wrap)
    add("alt_extreme", np.where(np.arange(64) % 2 == 0, 32767, -32768))   # -> 16
    add("all_m32768", np.full(64, -32768))
    for pos in (0, 7, 56, 63, 27):
        imp = np.zeros(64)
        imp[pos] = -255
        add("impulse_%d" % pos, imp)
    add("asymmetric", (np.arange(64) % 8) * 5 - (np.arange(64) // 8) * 11)
    return np.stack(blocks), names


def intra_refs_np(n, seed):
    """n reference sets (left[64] | top[65]): smooth ramps + noise, plus the WIP testbench's stimulus
    (src/mkIntra32-wip.bsv:539-543: left[i] = -(i+1), top[i] = i) as set 0 and flat / extreme sets."""
    r = splitmix64(seed, 0, n * 129)
    base = ((r >> np.uint64(17)) & np.uint64(0xFF)).astype(np.int64).reshape(n, 129)
    ramp = np.linspace(0, 255, 129)[None, :] * (((r.reshape(n, 129)[:, :1] >> np.uint64(40)) & np.uint64(3)).astype(np.int64) / 3.0)
    refs = np.clip(0.35 * base + 0.65 * ramp, 0, 255).astype(np.uint8)
    if n > 0:
        refs[0, :64] = (256 - (np.arange(64) + 1)) & 0xFF
        refs[0, 64:] = np.arange(65)
    if n > 1:
        refs[1] = 255
    if n > 2:
        refs[2] = 0
    if n > 3:
        refs[3] = np.where(np.arange(129) % 2 == 0, 255, 0)
    if n > 4:
        refs[4] = base[4].astype(np.uint8)                 # pure noise
    return refs
