"""Regression vectors for the stages upstream does not compute (tests/golden/frozen_unpinned.npz, made by
tests/golden/gen_frozen_golden.py from this repository's oracle).  They do not pin parity with the
reference -- there is nothing there to pin to -- they keep oracle and kernels from drifting."""
import os

import numpy as np
import pytest

from _util import GOLDEN_DIR

G = np.load(os.path.join(GOLDEN_DIR, "frozen_unpinned.npz"))


def test_oracle_reproduces_frozen_vectors(oracle):
    assert np.array_equal(oracle.dct32_inv(G["dct32_inv_in"]), G["dct32_inv_out"])
    for ttype, tname in ((0, "dct2"), (1, "dst7")):
        for n in (4, 8, 16):
            fwd = oracle.transform_fwd(ttype, n, G["%s_%d_in" % (tname, n)])
            assert np.array_equal(fwd, G["%s_%d_fwd" % (tname, n)]), (tname, n)
            assert np.array_equal(oracle.transform_inv(ttype, n, fwd), G["%s_%d_inv" % (tname, n)]), (tname, n)
    mv, cost, costs = oracle.satd_search(G["me_cur"], G["me_ref_padded"], int(G["me_pad"]), int(G["me_range"]), want_costs=True)
    assert np.array_equal(mv, G["me_mv"]) and np.array_equal(cost, G["me_cost"]) and np.array_equal(costs, G["me_costs"])
    modes = np.tile(np.arange(35, dtype=np.uint8), 6)
    idx = np.repeat(np.arange(6, dtype=np.uint32), 35)
    assert np.array_equal(oracle.intra32_predict(G["intra_refs"], modes, idx), G["intra_pred"])
    c, b = oracle.intra32_costs(G["intra_refs"], G["intra_src"])
    assert np.array_equal(c, G["intra_costs"]) and np.array_equal(b, G["intra_best"])


@pytest.mark.gpu
def test_kernels_reproduce_frozen_vectors():
    import x266_amd
    cd = x266_amd.Codec(0)
    assert np.array_equal(cd.dct32_inv(G["dct32_inv_in"]), G["dct32_inv_out"])
    for ttype, tname in ((0, "dct2"), (1, "dst7")):
        for n in (4, 8, 16):
            fwd = cd.transform_fwd(ttype, n, G["%s_%d_in" % (tname, n)])
            assert np.array_equal(fwd, G["%s_%d_fwd" % (tname, n)]), (tname, n)
            assert np.array_equal(cd.transform_inv(ttype, n, fwd), G["%s_%d_inv" % (tname, n)]), (tname, n)
    mv, cost, costs = cd.satd_search(G["me_cur"], G["me_ref_padded"], int(G["me_pad"]), int(G["me_range"]), want_costs=True)
    assert np.array_equal(mv, G["me_mv"]) and np.array_equal(cost, G["me_cost"]) and np.array_equal(costs, G["me_costs"])
    modes = np.tile(np.arange(35, dtype=np.uint8), 6)
    idx = np.repeat(np.arange(6, dtype=np.uint32), 35)
    assert np.array_equal(cd.intra32_predict(G["intra_refs"], modes, idx), G["intra_pred"])
    c, b = cd.intra32_costs(G["intra_refs"], G["intra_src"])
    assert np.array_equal(c, G["intra_costs"]) and np.array_equal(b, G["intra_best"])
