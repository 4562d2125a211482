"""CPU: the 32x32 intra predictor's oracle (oracle/intra_oracle.c, PARITY UNPINNED).
(a) its closed forms against the tables of the reference's WIP RTL sketch, read from
    /root/reference/src/mkIntra32-wip.bsv where that file is present (this container only);
(b) against an independent numpy statement of H.265 8.4.4.2.4-6;
(c) structural properties of the modes."""
import os
import re

import numpy as np
import pytest

from _util import intra_refs_np

BSV = "/root/reference/src/mkIntra32-wip.bsv"
ANGLES = [32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32]
INV = {11: -4096, 12: -1638, 13: -910, 14: -630, 15: -482, 16: -390, 17: -315, 18: -256, 19: -315, 20: -390, 21: -482,
       22: -630, 23: -910, 24: -1638, 25: -4096}


def test_angle_tables(oracle):
    L = oracle.lib
    assert [L.orc_intra_angle(m) for m in range(2, 35)] == ANGLES
    assert {m: L.orc_intra_inv_angle(m) for m in range(11, 26)} == INV
    for m, a in zip(range(11, 26), ANGLES[9:24]):            # invAngle = round(8192 / angle)
        assert abs(INV[m] - round(8192 / a)) <= 1


def _rows(text, name):
    body = text[text.index(name):]
    body = body[body.index("{") + 1:body.index("};")]
    return [[int(v) for v in re.findall(r"-?\d+", row)] for row in re.findall(r"\{([^{}]*)\}", body)]


@pytest.mark.skipif(not os.path.exists(BSV), reason="reference not present (GPU box)")
def test_closed_forms_match_the_reference_tables():
    text = open(BSV).read()
    map_tbl, fac_tbl = _rows(text, "Integer mapTbl"), _rows(text, "Integer facTbl")
    assert len(map_tbl) == 17 and len(fac_tbl) == 16 and all(len(r) == 32 for r in map_tbl + fac_tbl)
    k = np.arange(32) + 1
    typos = 0
    for row, mode in enumerate(range(2, 18)):                # rows 0..15 <-> HEVC modes 2..17 (the file's own comments)
        a = ANGLES[mode - 2]
        idx, fact = (k * a) >> 5, (k * a) & 31
        assert fac_tbl[row] == fact.tolist(), mode           # iFact, exactly
        want = (idx if a >= 0 else idx - idx.min()).tolist()  # iIdx; negative angles re-based to the smallest shift
        if map_tbl[row] != want:                             # the WIP file's "Mode 16" row has transposed entries
            assert mode == 16 and sorted(map_tbl[row]) == sorted(want)
            typos += int(np.sum(np.array(map_tbl[row]) != np.array(want)))
    assert typos == 2                                        # entries 22 and 23 of that row are swapped (4, 5 for 5, 4)
    assert map_tbl[16] == list(range(32, 0, -1))             # modes 18-25 row: pure diagonal walk
    # projected side references of getRefPixels (e.g. case 17 = HEVC mode 19): ((x*invAngle + 128) >> 8)
    case17 = text[text.index("        17: begin"):text.index("        18: begin")]
    listed = [int(v) for v in re.findall(r"xL\[\s*(\d+)\]", case17)]
    proj = [((x * INV[19] + 128) >> 8) for x in range(-1, ((32 * ANGLES[17]) >> 5) - 1, -1)]
    assert sorted(listed) == sorted(proj)[:-1]               # same projection; the file lists 25 of the 26 samples (stops at xL[31])


def _numpy_predict(left, top, mode):
    """H.265 8.4.4.2.4-6 for nTbS = 32, written independently of the C oracle (p[] as a dict)."""
    N = 32
    p = {(-1, -1): int(top[0])}
    for i in range(64):
        p[(i, -1)] = int(top[1 + i])
        p[(-1, i)] = int(left[i])
    out = np.zeros((N, N), np.int64)                         # out[y][x]
    if mode == 0:
        for y in range(N):
            for x in range(N):
                out[y, x] = ((N - 1 - x) * p[(-1, y)] + (x + 1) * p[(N, -1)] + (N - 1 - y) * p[(x, -1)] + (y + 1) * p[(-1, N)] + N) >> 6
        return out
    if mode == 1:
        out[:] = (sum(p[(x, -1)] for x in range(N)) + sum(p[(-1, y)] for y in range(N)) + N) >> 6
        return out
    a = ANGLES[mode - 2]
    ref = {}
    if mode >= 18:
        for x in range(0, N + 1):
            ref[x] = p[(-1 + x, -1)]
        if a < 0:
            if (N * a) >> 5 < -1:
                for x in range(-1, ((N * a) >> 5) - 1, -1):
                    ref[x] = p[(-1, -1 + ((x * INV[mode] + 128) >> 8))]
        else:
            for x in range(N + 1, 2 * N + 1):
                ref[x] = p[(-1 + x, -1)]
        for y in range(N):
            idx, f = ((y + 1) * a) >> 5, ((y + 1) * a) & 31
            for x in range(N):
                out[y, x] = ((32 - f) * ref[x + idx + 1] + f * ref[x + idx + 2] + 16) >> 5 if f else ref[x + idx + 1]
    else:
        for x in range(0, N + 1):
            ref[x] = p[(-1, -1 + x)]
        if a < 0:
            if (N * a) >> 5 < -1:
                for x in range(-1, ((N * a) >> 5) - 1, -1):
                    ref[x] = p[(-1 + ((x * INV[mode] + 128) >> 8), -1)]
        else:
            for x in range(N + 1, 2 * N + 1):
                ref[x] = p[(-1, -1 + x)]
        for x in range(N):
            idx, f = ((x + 1) * a) >> 5, ((x + 1) * a) & 31
            for y in range(N):
                out[y, x] = ((32 - f) * ref[y + idx + 1] + f * ref[y + idx + 2] + 16) >> 5 if f else ref[y + idx + 1]
    return out


def test_oracle_matches_numpy_statement(oracle):
    refs = intra_refs_np(7, 0x1357)
    modes = np.tile(np.arange(35, dtype=np.uint8), 7)
    idx = np.repeat(np.arange(7, dtype=np.uint32), 35)
    got = oracle.intra32_predict(refs, modes, idx)
    for i in range(modes.shape[0]):
        r = refs[idx[i]]
        assert np.array_equal(got[i].reshape(32, 32), _numpy_predict(r[:64], r[64:], int(modes[i]))), (idx[i], modes[i])


def test_mode_properties(oracle):
    refs = intra_refs_np(6, 99)
    r = refs[5]
    left, top = r[:64].astype(np.int64), r[64:].astype(np.int64)
    pred = lambda m: oracle.intra32_predict(r[None], np.array([m], np.uint8))[0].reshape(32, 32).astype(np.int64)
    assert np.array_equal(pred(26), np.tile(top[1:33], (32, 1)))                  # pure vertical copies the top row
    assert np.array_equal(pred(10), np.tile(left[:32, None], (1, 32)))            # pure horizontal copies the left column
    assert np.all(pred(1) == (top[1:33].sum() + left[:32].sum() + 32) >> 6)       # DC
    d34 = pred(34)                                                                # 45 degrees up-right: a walk along the top row
    assert all(np.array_equal(d34[y], top[2 + y:34 + y]) for y in range(32))
    d2 = pred(2)                                                                  # 45 degrees down-left: along the left column
    assert all(np.array_equal(d2[:, x], left[1 + x:33 + x]) for x in range(32))
    d18 = pred(18)                                                                # the main diagonal: corner on it
    assert all(d18[i, i] == top[0] for i in range(32)) and d18[0, 1] == top[1] and d18[1, 0] == left[0]
    flat = oracle.intra32_predict(refs[1][None].repeat(35, 0), np.arange(35, dtype=np.uint8))
    assert np.all(flat == 255)                                                    # every mode of a flat border is flat
    for m in range(2, 18):                                                        # transposing the border swaps the mode families
        rt = np.concatenate([top[1:65], [top[0]], left]).astype(np.uint8)
        a = oracle.intra32_predict(r[None], np.array([m], np.uint8))[0].reshape(32, 32)
        b = oracle.intra32_predict(rt[None], np.array([36 - m], np.uint8))[0].reshape(32, 32)
        assert np.array_equal(a, b.T), m
    with pytest.raises(AssertionError):
        oracle.intra32_predict(r[None], np.array([35], np.uint8))
