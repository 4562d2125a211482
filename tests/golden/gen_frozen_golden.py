#!/usr/bin/env python3
"""Freeze this repository's own definitions of the UNPINNED stages (nothing upstream computes them:
SURVEY.md section 8c) as small regression vectors, so that neither the oracle nor the kernels can
drift silently between rounds:

    python tests/golden/gen_frozen_golden.py     ->  tests/golden/frozen_unpinned.npz

Generated from oracle/liborc.so on deterministic inputs.  These are NOT reference outputs -- parity
for these stages stays "unpinned" -- they pin the repository to itself.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _util import Oracle, fullrange_np, intra_refs_np, me_frames, residual_np   # noqa: E402


def main():
    o = Oracle()
    out = {}
    z = np.concatenate([o.dct32_fwd(residual_np(6 * 1024, 0x266)), fullrange_np(2 * 1024, 5).reshape(2, 1024)])
    out["dct32_inv_in"], out["dct32_inv_out"] = z, o.dct32_inv(z)
    for ttype, tname in ((0, "dct2"), (1, "dst7")):
        for n in (4, 8, 16):
            x = np.concatenate([residual_np(24 * n * n, 40 + n + ttype), fullrange_np(8 * n * n, 50 + n + ttype)]).reshape(-1, n * n)
            out["%s_%d_in" % (tname, n)] = x
            out["%s_%d_fwd" % (tname, n)] = o.transform_fwd(ttype, n, x)
            out["%s_%d_inv" % (tname, n)] = o.transform_inv(ttype, n, out["%s_%d_fwd" % (tname, n)])
    cur, refp = me_frames(48, 32, 8, 4242, mv=(2, -1))
    mv, cost, costs = o.satd_search(cur, refp, 8, 6, want_costs=True)
    out.update(me_cur=cur, me_ref_padded=refp, me_pad=np.array(8), me_range=np.array(6), me_mv=mv, me_cost=cost, me_costs=costs)
    refs = intra_refs_np(6, 0x1A7)
    modes = np.tile(np.arange(35, dtype=np.uint8), 6)
    idx = np.repeat(np.arange(6, dtype=np.uint32), 35)
    pred = o.intra32_predict(refs, modes, idx)
    src = pred[np.array([7, 35 + 1, 70 + 26, 105 + 10, 140 + 18, 175 + 30])].copy()
    src[:, ::3] ^= 3
    c, b = o.intra32_costs(refs, src)
    out.update(intra_refs=refs, intra_pred=pred, intra_src=src, intra_costs=c, intra_best=b)
    np.savez_compressed(os.path.join(HERE, "frozen_unpinned.npz"), **out)
    print("wrote frozen_unpinned.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
