"""CPU: the oracle against the real reference compiled in place (oracle/_ref).
Skipped where the reference build is absent; the committed golden vectors
(test_oracle_golden.py) carry the same pin everywhere else."""
import numpy as np

from _util import extremes_np, fullrange_np, residual_np


def _mixed(n_each, unit, seed):
    return np.concatenate([
        residual_np(n_each * unit, seed).reshape(n_each, unit),
        fullrange_np(n_each * unit, seed + 1).reshape(n_each, unit),
        extremes_np(n_each * unit, seed + 2).reshape(n_each, unit)])


def test_table_identical(oracle, reference):
    assert np.array_equal(oracle.table(), reference.table())


def test_dct32_fwd_identical(oracle, reference):
    x = _mixed(3000, 1024, 11)
    assert np.array_equal(oracle.dct32_fwd(x, threads=4), reference.dct32_fwd(x))


def test_dct32_single_pass_identical(oracle, reference):
    x = _mixed(40, 1024, 21)
    for blk in x:
        for shift in (4, 11):
            assert np.array_equal(oracle.dct32_pass(blk, shift), reference.dct32_pass(blk, shift))


def test_satd_identical(oracle, reference):
    d = _mixed(100000, 64, 31)
    assert np.array_equal(oracle.satd8x8(d, threads=4), reference.satd8x8(d))
