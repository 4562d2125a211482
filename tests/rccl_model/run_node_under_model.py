#!/usr/bin/env python3
"""Run in its own process with X266HIP_RCCL_LIB = the RCCL model: the node layer's RCCL transport with N ranks on one GPU.
Compares every result with the single-device calls of the same library (themselves checked against the oracle
elsewhere).  Prints 'ok <what>' lines; exit code 0 only if everything matched and the model saw no unmatched transfer."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd.node import Node, OP_DCT32_FWD, OP_SATD8X8

world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
model = ctypes.CDLL(os.environ["X266HIP_RCCL_LIB"])
node = Node.single_process([0] * world)
codec = x266_amd.Codec(0)
node.self_test()                                            # ring send/recv + all-reduce through the model
print("ok self_test")


def dev(arr):
    b = codec.alloc(max(arr.nbytes, 16))
    b.upload(arr)
    return b


rs = np.random.RandomState(7)
for (w, h, n_frames) in ((96, 160, 6), (64, 32, 4), (1920, 1088, 3)):
    nd, ns = (w // 32) * (h // 32), (w // 8) * (h // 8)
    st = node.frame_stream(w, h)
    frames = [(rs.randint(-255, 256, nd * 1024).astype(np.int16), rs.randint(-255, 256, ns * 64).astype(np.int16)) for _ in range(n_frames)]
    bufs = [(dev(a), dev(b), codec.alloc(nd * 2048), codec.alloc(ns * 4)) for a, b in frames]
    for (da, db, dc, de) in bufs:
        st.push([da.ptr, db.ptr], [dc.ptr, de.ptr])
    st.flush()
    for (a, b), (da, db, dc, de) in zip(frames, bufs):
        assert np.array_equal(dc.download(np.int16, nd * 1024), codec.dct32_fwd(a).ravel()), (w, h)
        assert np.array_equal(de.download(np.uint32, ns), codec.satd8x8(b)), (w, h)
    st.close()
    print("ok frame stream %dx%d, %d frames, %d ranks" % (w, h, n_frames, world))

# unit counts that change from frame to frame (zero, one, fewer than ranks): empty transfers are skipped on both ends
from x266_amd.node import OP_DCT32_INV
caps = [19, 301]
st = node.stream([OP_DCT32_INV, OP_SATD8X8], caps)
zin = rs.randint(-255, 256, caps[0] * 1024).astype(np.int16)
din = rs.randint(-255, 256, caps[1] * 64).astype(np.int16)
dz, dd = dev(zin), dev(din)
plans = [[19, 301], [0, 0], [1, 1], [world - 1, world + 1], [0, 5], [7, 0], [18, 300]]
outs = [(codec.alloc(caps[0] * 2048), codec.alloc(caps[1] * 4)) for _ in plans]
for units, (o0, o1) in zip(plans, outs):
    st.push([dz.ptr, dd.ptr], [o0.ptr, o1.ptr], units)
st.flush()
for units, (o0, o1) in zip(plans, outs):
    a, b = units
    assert np.array_equal(o0.download(np.int16, caps[0] * 1024)[: a * 1024], codec.dct32_inv(zin[: a * 1024]).ravel()), units
    assert np.array_equal(o1.download(np.uint32, caps[1])[:b], codec.satd8x8(din[: b * 64])), units
st.close()
print("ok ragged unit counts")

n = 5003
x = rs.randint(-255, 256, n * 1024).astype(np.int16)
di, do = dev(x), codec.alloc(n * 2048)
node.batch_scatter_gather(OP_DCT32_FWD, di.ptr, do.ptr, n, 700)
assert np.array_equal(do.download(np.int16, n * 1024), codec.dct32_fwd(x).ravel())
print("ok batch scatter-gather")

w, h, rng = 200, 136, 24
cur = rs.randint(0, 256, (h, w)).astype(np.uint8)
refp = rs.randint(0, 256, (h + 2 * rng, w + 2 * rng)).astype(np.uint8)
mv0, cost0, _ = codec.satd_search(cur, refp, rng, rng)
dc, dr = dev(cur), dev(refp)
nb = (h // 8) * (w // 8)
for n_stripes in (0, world + 2):
    db = codec.alloc(nb * 8)
    node.satd_search(dc.ptr, w, dr.ptr + rng * (w + 2 * rng) + rng, w + 2 * rng, w, h, rng, n_stripes, db.ptr)
    raw = db.download(np.uint8, nb * 8)
    assert np.array_equal(raw.view(np.int16).reshape(nb, 4)[:, :2], mv0) and np.array_equal(raw.view(np.uint32).reshape(nb, 2)[:, 1], cost0)
print("ok sharded motion search")
model.rccl_model_errors.restype = ctypes.c_int
assert model.rccl_model_errors() == 0
node.close()
# the model is not a rubber stamp: an unmatched send and a size mismatch are refused
P = ctypes.c_void_p
comms = (P * 2)()
assert model.ncclCommInitAll(comms, 2, (ctypes.c_int * 2)(0, 0)) == 0
buf = codec.alloc(64)
model.ncclSend.argtypes = [P, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, P, P]
model.ncclRecv.argtypes = [P, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, P, P]
model.ncclGroupStart()
assert model.ncclSend(buf.ptr, 16, 1, 1, comms[0], None) == 0
assert model.ncclGroupEnd() != 0                                   # a send nobody receives: a hang with real RCCL
model.ncclGroupStart()
assert model.ncclSend(buf.ptr, 16, 1, 1, comms[0], None) == 0
assert model.ncclRecv(buf.ptr + 32, 8, 1, 0, comms[1], None) == 0
assert model.ncclGroupEnd() != 0                                   # 16 bytes sent, 8 expected
assert model.rccl_model_errors() == 2
print("ok all")
