#!/usr/bin/env python3
"""Generate the committed golden vectors from the REAL reference.

Run in the build container only (needs /root/reference):
    make -C oracle ref && python tests/golden/gen_golden.py

The reference holds no stored vectors for this path (the
Bluesim testbenches compare against the C model at run time), so the fixtures
are outputs of the reference's own golden models (src_tb/dct32.c, satd.c,
compiled in place as oracle/_ref/libx266ref.so) on deterministic inputs:

  dct32_fwd.npz   inputs  [N,1024] int16 + outputs [N,1024] int16 + names
  satd8x8.npz     inputs  [N,64]   int16 + outputs [N] uint32     + names
  bdpi_dct32.npz  the BDPI call sequence of src/mkDct32.bsv:430-470 for the
                  11 dct32_genNew() blocks the testbench runs before $finish
                  (passed 0..10, src/mkDct32.bsv:472-478) in a fresh process
                  (glibc rand(), default seed): getDiff words [11,16,32]
                  uint32, getDct words [11,256] uint64, plus the raw blocks
  bdpi_satd.npz   same for src/mkSatd.bsv:215-252 at the testbench's own
                  length, 256 blocks (cnt 0..255, src/mkSatd.bsv:235-252)

Fixtures are data (inputs + expected outputs); no reference source is stored.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _util import (Reference, dct_edge_blocks, extremes_np, fullrange_np,   # noqa: E402
                   residual_np, satd_edge_blocks)

DCT_SEED, SATD_SEED = 0x266, 0x267


def gen_dct(ref):
    tri = residual_np(24 * 1024, DCT_SEED).reshape(24, 1024)          # blocks 0..23 of the bench stream
    full = fullrange_np(16 * 1024, DCT_SEED + 0x1000).reshape(16, 1024)
    ext = extremes_np(8 * 1024, DCT_SEED + 0x2000).reshape(8, 1024)
    edge, names = dct_edge_blocks()
    x = np.concatenate([tri, full, ext, edge]).astype(np.int16)
    names = ["stream_%d" % i for i in range(24)] + ["fullrange_%d" % i for i in range(16)] + \
            ["extreme_%d" % i for i in range(8)] + names
    y = ref.dct32_fwd(x)
    # intermediate of the first pass for block 0 (pins shift 4 + transposed store)
    p1 = ref.dct32_pass(x[0], 4)
    np.savez_compressed(os.path.join(HERE, "dct32_fwd.npz"), inputs=x, outputs=y,
                        names=np.array(names), pass1_block0=p1)
    print("dct32_fwd.npz:", x.shape, "blocks")


def gen_satd(ref):
    tri = residual_np(1536 * 64, SATD_SEED).reshape(1536, 64)
    full = fullrange_np(384 * 64, SATD_SEED + 0x1000).reshape(384, 64)
    ext = extremes_np(116 * 64, SATD_SEED + 0x2000).reshape(116, 64)
    edge, names = satd_edge_blocks()
    x = np.concatenate([tri, full, ext, edge]).astype(np.int16)
    names = ["stream"] * 1536 + ["fullrange"] * 384 + ["extreme"] * 116 + names
    y = ref.satd8x8(x)
    np.savez_compressed(os.path.join(HERE, "satd8x8.npz"), inputs=x, outputs=y, names=np.array(names))
    print("satd8x8.npz:", x.shape, "blocks; edge results:", dict(zip(names[-len(edge):], y[-len(edge):].tolist())))


TB_DCT_BLOCKS, TB_SATD_BLOCKS = 11, 256          # what mkTb runs before $finish (src/mkDct32.bsv:472-478, src/mkSatd.bsv:235-252)


def bdpi_dct_child():
    ref = Reference()
    L = ref.lib
    n = TB_DCT_BLOCKS
    diffs = np.zeros((n, 16, 32), np.uint32)
    words = np.zeros((n, 256), np.uint64)
    mats = np.zeros((n, 1024), np.int16)
    dcts = np.zeros((n, 1024), np.int16)
    for b in range(n):
        L.dct32_genNew()
        mats[b] = np.ctypeslib.as_array(L.ref_dct32_last_input(), (1024,))
        dcts[b] = np.ctypeslib.as_array(L.ref_dct32_last_output(), (1024,))
        res = (ctypes.c_uint32 * 32)()
        for i in range(16):
            L.dct32_getDiff(res)
            diffs[b, i] = np.frombuffer(res, np.uint32)
        for i in range(256):
            words[b, i] = L.dct32_getDct()
    np.savez_compressed(os.path.join(HERE, "bdpi_dct32.npz"), diff_words=diffs, dct_words=words,
                        blocks=mats, dcts=dcts)
    print("bdpi_dct32.npz: first word %08X, first dct word %016X" % (diffs[0, 0, 0], words[0, 0]))


def bdpi_satd_child():
    ref = Reference()
    L = ref.lib
    n = TB_SATD_BLOCKS
    diffs = np.zeros((n, 8, 4), np.uint32)
    satd = np.zeros(n, np.uint32)
    mats = np.zeros((n, 64), np.int16)
    for b in range(n):
        L.satd8x8_genNew()
        mats[b] = np.ctypeslib.as_array(L.ref_satd8x8_last_input(), (64,))
        res = (ctypes.c_uint32 * 4)()
        for i in range(8):
            L.satd8x8_getDiff(res)
            diffs[b, i] = np.frombuffer(res, np.uint32)
        satd[b] = L.satd8x8_getSatd()
    np.savez_compressed(os.path.join(HERE, "bdpi_satd.npz"), diff_words=diffs, satd=satd, blocks=mats)
    print("bdpi_satd.npz: %d blocks, satd" % n, satd[:8].tolist(), "...")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--bdpi-dct":
        bdpi_dct_child()
    elif len(sys.argv) > 1 and sys.argv[1] == "--bdpi-satd":
        bdpi_satd_child()
    else:
        ref = Reference()
        gen_dct(ref)
        gen_satd(ref)
        # rand() is process-global state: each BDPI fixture comes from a fresh process
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--bdpi-dct"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--bdpi-satd"])
