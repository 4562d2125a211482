"""ctypes view of the node part of libx266hip.so (include/x266hip.h, "one node, several GPUs"):
a binding, nothing else -- the scatter -> transform -> gather schedule, the RCCL groups and the
stripe plan all live in x266_amd/csrc/x266hip_node.cpp.

Two ways to make a node, as in C:
    Node.single_process(devices)          one process drives n GPUs       (xHipNodeInit)
    Node.for_rank(device, rank, world, id) one process per GPU             (xHipNodeInitRank);
        `id` = Node.unique_id() on rank 0, handed to every rank by the host (e.g. a
        torch.distributed broadcast -- the only thing the control plane is needed for).
"""
import ctypes

from ._lib import X266Error, load_library

_P = ctypes.c_void_p
_SZ = ctypes.c_size_t
NODE_ID_BYTES = 128
OP_DCT32_FWD, OP_DCT32_INV, OP_SATD8X8 = 0, 1, 2
_protos_done = False


def _lib():
    global _protos_done
    L = load_library()
    if _protos_done:
        return L
    L.xShardRange.argtypes = [_SZ, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_SZ), ctypes.POINTER(_SZ)]
    L.xMeStripePlan.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_int)] * 4
    L.xHipNodeInit.argtypes = [ctypes.POINTER(_P), _P, ctypes.c_int]
    L.xHipNodeUniqueId.argtypes = [_P]
    L.xHipNodeInitRank.argtypes = [ctypes.POINTER(_P), ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]
    L.xHipNodeFree.argtypes = [_P]
    L.xHipNodeFree.restype = None
    L.xHipNodeWorld.argtypes = [_P]
    L.xHipNodeLocalCount.argtypes = [_P]
    L.xHipNodeLocalRank.argtypes = [_P, ctypes.c_int]
    L.xHipNodeCtx.argtypes = [_P, ctypes.c_int]
    L.xHipNodeCtx.restype = _P
    L.xHipNodeLastError.argtypes = [_P]
    L.xHipNodeLastError.restype = ctypes.c_char_p
    L.xHipNodeSetOption.argtypes = [_P, ctypes.c_char_p, ctypes.c_int]
    L.xHipNodeSelfTest.argtypes = [_P]
    L.xHipNodeRcclInfo.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, _SZ]
    L.xNodeStreamNextSlotStream.argtypes = [_P]
    L.xNodeStreamNextSlotStream.restype = _P
    L.xNodeStreamCreate.argtypes = [_P, ctypes.c_int, _P, _P, ctypes.POINTER(_P)]
    L.xNodeFrameStreamCreate.argtypes = [_P, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_P)]
    L.xNodeStreamFree.argtypes = [_P]
    L.xNodeStreamFree.restype = None
    L.xNodeStreamPush.argtypes = [_P, _P, _P, _P, _P, ctypes.POINTER(ctypes.c_long)]
    L.xNodeStreamFlush.argtypes = [_P]
    L.xNodeStreamWait.argtypes = [_P, ctypes.c_long]
    L.xNodeBatchScatterGather.argtypes = [_P, ctypes.c_int, _P, _P, _SZ, _SZ]
    L.xNodeSatd8x8Search.argtypes = [_P, _P, ctypes.c_ssize_t, _P, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, _P]
    _protos_done = True
    return L


def shard_range(n_units, rank, world):
    """xShardRange: rank's contiguous [begin, end) of n_units."""
    b, e = _SZ(), _SZ()
    if _lib().xShardRange(n_units, rank, world, ctypes.byref(b), ctypes.byref(e)) != 0:
        raise ValueError("bad rank/world: %d/%d" % (rank, world))
    return b.value, e.value


def me_stripe_plan(height, rng, stripe, n_stripes):
    """xMeStripePlan -> ((block_row_begin, block_row_end), (ref_row_begin, ref_row_end))."""
    v = [ctypes.c_int() for _ in range(4)]
    if _lib().xMeStripePlan(height, rng, stripe, n_stripes, *[ctypes.byref(x) for x in v]) != 0:
        raise ValueError("bad stripe plan arguments")
    return (v[0].value, v[1].value), (v[2].value, v[3].value)


class Node:
    def __init__(self, handle):
        self.L = _lib()
        self.h = handle

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(NODE_ID_BYTES)
        rc = _lib().xHipNodeUniqueId(buf)
        if rc != 0:
            raise X266Error("xHipNodeUniqueId failed (%d): RCCL could not be loaded" % rc)
        return buf.raw

    @classmethod
    def single_process(cls, devices):
        devices = list(devices)
        arr = (ctypes.c_int * len(devices))(*devices)
        h = _P()
        rc = _lib().xHipNodeInit(ctypes.byref(h), arr, len(devices))
        if rc != 0 or not h.value:
            raise X266Error("xHipNodeInit(%r) failed with %d" % (devices, rc))
        return cls(h)

    @classmethod
    def for_rank(cls, device, rank, world, uid):
        assert len(uid) == NODE_ID_BYTES
        h = _P()
        rc = _lib().xHipNodeInitRank(ctypes.byref(h), device, rank, world, ctypes.c_char_p(uid))
        if rc != 0 or not h.value:
            raise X266Error("xHipNodeInitRank(device %d, rank %d of %d) failed with %d" % (device, rank, world, rc))
        return cls(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.xHipNodeFree(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise X266Error("%s failed (%d): %s" % (what, rc, self.L.xHipNodeLastError(self.h).decode()))

    @property
    def world(self):
        return self.L.xHipNodeWorld(self.h)

    @property
    def local_ranks(self):
        return [self.L.xHipNodeLocalRank(self.h, i) for i in range(self.L.xHipNodeLocalCount(self.h))]

    @property
    def drives_root(self):
        return 0 in self.local_ranks

    def rank_codec(self, local_index):
        """The context of the local_index-th rank this process drives (xHipNodeCtx), as a borrowed Codec: lets a host run its own
        work on that rank's device through the same ABI."""
        from ._lib import Codec
        p = self.L.xHipNodeCtx(self.h, local_index)
        if not p:
            raise X266Error("xHipNodeCtx(%d): no such local rank" % local_index)
        return Codec.borrowed(p)

    def set_option(self, key, value):
        self._check(self.L.xHipNodeSetOption(self.h, key.encode(), int(value)), "xHipNodeSetOption(%s)" % key)

    def self_test(self):
        self._check(self.L.xHipNodeSelfTest(self.h), "xHipNodeSelfTest")

    @staticmethod
    def rccl_info():
        """(version, path) of the RCCL this process loaded for the node layer; (0, "") when none could be opened."""
        v = ctypes.c_int()
        buf = ctypes.create_string_buffer(1024)
        if _lib().xHipNodeRcclInfo(ctypes.byref(v), buf, 1024) != 0:
            return 0, ""
        return v.value, buf.value.decode()

    def frame_stream(self, width, height):
        s = _P()
        self._check(self.L.xNodeFrameStreamCreate(self.h, width, height, ctypes.byref(s)), "xNodeFrameStreamCreate")
        return NodeStream(self, s, 2)

    def stream(self, ops, max_units):
        n = len(ops)
        a_ops = (ctypes.c_int * n)(*ops)
        a_units = (_SZ * n)(*max_units)
        s = _P()
        self._check(self.L.xNodeStreamCreate(self.h, n, a_ops, a_units, ctypes.byref(s)), "xNodeStreamCreate")
        return NodeStream(self, s, n)

    def batch_scatter_gather(self, op, d_in, d_out, n_units, chunk_units=0):
        self._check(self.L.xNodeBatchScatterGather(self.h, op, d_in or None, d_out or None, n_units, chunk_units),
                    "xNodeBatchScatterGather")

    def satd_search(self, d_cur, cur_stride, d_ref_origin, ref_stride, width, height, rng, n_stripes, d_best):
        self._check(self.L.xNodeSatd8x8Search(self.h, d_cur or None, cur_stride, d_ref_origin or None, ref_stride, width, height,
                                              rng, n_stripes, d_best or None), "xNodeSatd8x8Search")


class NodeStream:
    def __init__(self, node, handle, n_lanes):
        self.node, self.s, self.n_lanes = node, handle, n_lanes

    def push(self, d_in=None, d_out=None, units=None, producer_stream=0):
        """d_in / d_out: per-lane device pointers (ints) on the process driving rank 0, None elsewhere."""
        n = self.n_lanes
        a_in = (_P * n)(*d_in) if d_in is not None else None
        a_out = (_P * n)(*d_out) if d_out is not None else None
        a_units = (_SZ * n)(*units) if units is not None else None
        t = ctypes.c_long()
        self.node._check(self.node.L.xNodeStreamPush(self.s, a_in, a_out, a_units, producer_stream or None, ctypes.byref(t)),
                         "xNodeStreamPush")
        return t.value

    def prepare(self, d_in, d_out, units=None):
        """The ctypes argument arrays of one push, built once: hosts that cycle through a ring of frame buffers (bench.py) keep
        one per ring entry and hand it to push_prepared, so that a frame costs one foreign call and nothing else."""
        n = self.n_lanes
        return ((_P * n)(*d_in), (_P * n)(*d_out), (_SZ * n)(*units) if units is not None else None)

    def push_prepared(self, prepared, producer_stream=0):
        rc = self.node.L.xNodeStreamPush(self.s, prepared[0], prepared[1], prepared[2], producer_stream or None, None)
        if rc:
            self.node._check(rc, "xNodeStreamPush")

    def next_slot_stream(self):
        """The root-device stream the next pushed frame's kernels run on (0 where this process does not drive the root)."""
        return self.node.L.xNodeStreamNextSlotStream(self.s) or 0

    def flush(self):
        self.node._check(self.node.L.xNodeStreamFlush(self.s), "xNodeStreamFlush")

    def wait(self, ticket):
        self.node._check(self.node.L.xNodeStreamWait(self.s, ticket), "xNodeStreamWait")

    def close(self):
        if getattr(self, "s", None):
            self.node.L.xNodeStreamFree(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
