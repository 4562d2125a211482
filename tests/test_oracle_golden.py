"""CPU: the oracle (oracle/*.c) against the committed golden vectors, which are
outputs of the REAL reference (src_tb/dct32.c, satd.c) -- see
tests/golden/gen_golden.py.  This is what pins the oracle."""
import os

import numpy as np

from _util import GOLDEN_DIR, dct_edge_blocks, residual_np, satd_edge_blocks


def _load(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def test_dct32_fwd_matches_golden(oracle):
    g = _load("dct32_fwd.npz")
    out = oracle.dct32_fwd(g["inputs"])
    assert np.array_equal(out, g["outputs"])


def test_dct32_first_pass_matches_golden(oracle):
    g = _load("dct32_fwd.npz")
    p1 = oracle.dct32_pass(g["inputs"][0], 4)
    assert np.array_equal(p1, g["pass1_block0"])
    # and the dense contraction is the same integer
    assert np.array_equal(oracle.dct32_pass(g["inputs"][0], 4, dense=True), g["pass1_block0"])


def test_golden_inputs_are_reproducible(oracle):
    """The first 24 golden blocks are blocks 0..23 of the bench stream (seed 0x266)."""
    g = _load("dct32_fwd.npz")
    assert np.array_equal(g["inputs"][:24].ravel(), oracle.fill_residual(24 * 1024, 0x266))
    assert np.array_equal(g["inputs"][:24].ravel(), residual_np(24 * 1024, 0x266))
    edge, names = dct_edge_blocks()
    assert np.array_equal(g["inputs"][48:], edge)
    assert list(g["names"][48:]) == names


def test_satd_matches_golden(oracle):
    g = _load("satd8x8.npz")
    assert np.array_equal(oracle.satd8x8(g["inputs"]), g["outputs"])


def test_satd_known_answers(oracle):
    """Input-independent answers recorded from the real reference."""
    edge, names = satd_edge_blocks()
    got = dict(zip(names, oracle.satd8x8(edge).tolist()))
    assert got["zeros"] == 0
    assert got["all_255"] == 4080
    assert got["all_m256"] == 4096
    assert got["all_32767"] == 16          # int16 wraparound must be reproduced
    assert got["alt_extreme"] == 16
    g = _load("satd8x8.npz")
    assert np.array_equal(g["inputs"][-len(edge):], edge)


def test_bdpi_dct_sequence(oracle):
    """The BDPI word packing of the oracle on the reference's own stimulus
    (glibc rand(), default seed) reproduces the reference's call sequence."""
    g = _load("bdpi_dct32.npz")
    assert int(g["diff_words"][0, 0, 0]) == 0xFFF6FFA1          # SURVEY.md 8c known answers
    assert int(g["diff_words"][0, 0, 16]) == 0xFF9DFF87
    assert int(g["dct_words"][0, 0]) == 0xFFF70017FDBAFF87
    assert int(g["dct_words"][0, 1]) == 0xFEDCFEE0FEB3FD53
    for b in range(g["blocks"].shape[0]):
        mat = g["blocks"][b]
        dct = oracle.dct32_fwd(mat)[0]
        assert np.array_equal(dct, g["dcts"][b])
        for i in range(16):
            assert np.array_equal(oracle.pack_diff_rows(mat, 2 * i), g["diff_words"][b, i])
        words = [oracle.pack_dct_word(dct, 4 * i) for i in range(256)]
        assert words == [int(w) for w in g["dct_words"][b]]


def test_bdpi_satd_sequence(oracle):
    g = _load("bdpi_satd.npz")
    assert g["satd"][:3].tolist() == [10867, 10533, 11552]
    assert np.array_equal(oracle.satd8x8(g["blocks"]), g["satd"])
    words = g["blocks"].astype(np.int16).view(np.uint32).reshape(-1, 8, 4)
    assert np.array_equal(words, g["diff_words"])


def test_sad_known_answer_fixture():
    """The reference's only stored known answer: 64x64 SAD = 344807
    (riscv/programs/benchmarks/sad/dataset1.h:423-426, checked by sad.c:57)."""
    g = _load("sad64.npz")
    assert int(np.abs(g["a"].astype(np.int32) - g["b"].astype(np.int32)).sum()) == int(g["sad"][0]) == 344807
