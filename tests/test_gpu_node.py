"""GPU: the node part of the C ABI (include/x266hip.h "one node, several GPUs") -- BASELINE configs[4].

On the one-GPU test box a node has one rank: the RCCL communicator, the send/recv groups and the
all-reduce still execute (xHipNodeSelfTest sends to / receives from itself inside a group), the pipelined
schedule runs with its streams and events, and the frame is transformed in place.  What cannot run here is
a transfer between two devices; its plan (shards, stripes, ordering) is covered by the gloo tests and by
host/stream8k.c, which validates any number of devices against the single-device result."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import x266_amd
from x266_amd.node import Node, OP_DCT32_FWD, OP_DCT32_INV, OP_SATD8X8
from _dev import Dev, hip_runtime
from _util import me_frames

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Lazy:
    """device memory on device 0 through the C ABI (tests/_dev.py); the context is created with the first allocation"""
    _d = None

    def __getattr__(self, name):
        if _Lazy._d is None:
            _Lazy._d = Dev(x266_amd.Codec(0))
        return getattr(_Lazy._d, name)


D = _Lazy()


@pytest.fixture(scope="module")
def node():
    n = Node.single_process([0])
    yield n
    n.close()


def test_rccl_executes_on_this_box(node):
    assert node.world == 1 and node.local_ranks == [0] and node.drives_root
    node.self_test()                                  # ncclCommInitAll + group{send, recv} + all-reduce, checked word by word


def test_process_per_gpu_init_with_unique_id():
    uid = Node.unique_id()
    n = Node.for_rank(0, 0, 1, uid)                   # the call sequence bench.py --gpus N makes on every rank
    n.self_test()
    n.close()


def _frame(oracle, geo_blocks, seed, f):
    return oracle.fill_residual(geo_blocks, seed, f * 10 ** 7)


@pytest.mark.parametrize("w,h,n_frames", [(96, 160, 7), (7680, 4320, 4)])
def test_frame_stream_is_bit_exact(node, oracle, w, h, n_frames):
    """configs[4] at its own size: 7680x4320 = 32 400 DCT32 + 518 400 SATD blocks per frame, pipelined
    (two frames in flight, slots reused), every frame compared whole with the oracle."""
    n_d, n_s = (w // 32) * (h // 32), (w // 8) * (h // 8)
    st = node.frame_stream(w, h)
    xin = [(D.from_numpy(_frame(oracle, n_d * 1024, 0x266, f)), D.from_numpy(_frame(oracle, n_s * 64, 0x267, f)))
           for f in range(n_frames)]
    out = [(D.zeros(n_d * 1024, np.int16), D.zeros(n_s, np.int32)) for _ in range(n_frames)]
    D.synchronize()
    tickets = []
    for f in range(n_frames):
        tickets.append(st.push([xin[f][0].data_ptr(), xin[f][1].data_ptr()], [out[f][0].data_ptr(), out[f][1].data_ptr()]))
        if f >= 2:
            st.wait(tickets[f - 2])                   # complete two steps later, without a flush
            assert np.array_equal(out[f - 2][0].cpu().numpy(), oracle.dct32_fwd(xin[f - 2][0].cpu().numpy(), threads=32).ravel()), f - 2
    with pytest.raises(x266_amd.X266Error):
        st.wait(tickets[-1])                          # its results have not travelled yet
    st.flush()
    for f in range(n_frames):
        assert np.array_equal(out[f][0].cpu().numpy(), oracle.dct32_fwd(xin[f][0].cpu().numpy(), threads=32).ravel()), f
        assert np.array_equal(out[f][1].cpu().numpy(), oracle.satd8x8(xin[f][1].cpu().numpy(), threads=32).astype(np.int32)), f
    st.close()


def test_output_buffer_of_a_frame_in_flight_is_refused(node, oracle):
    """buffer ownership is checked, not only documented -- an output buffer that overlaps the output of one of the previous
    X266_STREAM_OUT_RING - 1 frames still in flight is EINVAL (inputs may be shared: they are only read); a flush or a waited ticket
    gives the buffers back."""
    st = node.stream([OP_DCT32_FWD], [8])
    x = D.from_numpy(oracle.fill_residual(8 * 1024, 3))
    outs = [D.zeros(8 * 1024, np.int16) for _ in range(5)]
    D.synchronize()
    t0 = st.push([x.data_ptr()], [outs[0].data_ptr()])
    for back in (1, 2, 3, 4):                                            # frame `back` steps later, same output: refused every time
        with pytest.raises(x266_amd.X266Error, match="in flight"):
            st.push([x.data_ptr()], [outs[0].data_ptr() + 2048 * 3], [5])   # a partial overlap counts
        st.push([x.data_ptr()], [outs[back].data_ptr()])                 # the same INPUT is fine
    st.push([x.data_ptr()], [outs[0].data_ptr()])                        # five frames later: legal
    st.wait(t0 + 3)
    st.push([x.data_ptr()], [outs[1].data_ptr()])                        # frames <= t0 + 3 are complete: their outputs may be reused at once
    st.flush()
    want = oracle.dct32_fwd(x.cpu().numpy()).ravel()
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), want)
    st.push([x.data_ptr()], [outs[0].data_ptr()])                        # after a flush nothing is in flight
    st.push([x.data_ptr()], [outs[1].data_ptr()])
    st.flush()
    st.close()


def test_stream_takes_ragged_unit_counts_and_inverse_lane(node, oracle):
    st = node.stream([OP_DCT32_INV, OP_SATD8X8, OP_DCT32_FWD], [40, 1000, 40])
    z = D.from_numpy(oracle.fill_residual(40 * 1024, 5))
    d = D.from_numpy(oracle.fill_residual(1000 * 64, 6))
    for units in ([40, 1000, 40], [17, 999, 0], [1, 1, 40]):
        o0 = D.zeros(40 * 1024, np.int16)
        o1 = D.zeros(1000, np.int32)
        o2 = D.zeros(40 * 1024, np.int16)
        D.synchronize()
        st.push([z.data_ptr(), d.data_ptr(), z.data_ptr()], [o0.data_ptr(), o1.data_ptr(), o2.data_ptr()], units)
        st.flush()
        a, b, c = units
        assert np.array_equal(o0.cpu().numpy()[: a * 1024], oracle.dct32_inv(z.cpu().numpy()[: a * 1024]).ravel())
        assert not o0.cpu().numpy()[a * 1024:].any()
        assert np.array_equal(o1.cpu().numpy()[:b], oracle.satd8x8(d.cpu().numpy()[: b * 64]).astype(np.int32))
        assert np.array_equal(o2.cpu().numpy()[: c * 1024], oracle.dct32_fwd(z.cpu().numpy()[: c * 1024]).ravel())
    with pytest.raises(x266_amd.X266Error):
        st.push([z.data_ptr(), d.data_ptr(), z.data_ptr()], [o0.data_ptr(), o1.data_ptr(), o2.data_ptr()], [41, 1, 1])
    st.close()


@pytest.mark.parametrize("op,n,chunk", [(OP_DCT32_FWD, 10000, 0), (OP_SATD8X8, 200001, 4096), (OP_DCT32_INV, 5, 2)])
def test_batch_scatter_gather(node, oracle, op, n, chunk):
    unit = 64 if op == OP_SATD8X8 else 1024
    x = oracle.fill_residual(n * unit, 77)
    tin = D.from_numpy(x)
    tout = D.zeros(n if op == OP_SATD8X8 else n * 1024, np.int32 if op == OP_SATD8X8 else np.int16)
    D.synchronize()
    node.batch_scatter_gather(op, tin.data_ptr(), tout.data_ptr(), n, chunk)
    want = {OP_DCT32_FWD: lambda: oracle.dct32_fwd(x, threads=16).ravel(), OP_DCT32_INV: lambda: oracle.dct32_inv(x, threads=16).ravel(),
            OP_SATD8X8: lambda: oracle.satd8x8(x, threads=16).astype(np.int32)}[op]()
    assert np.array_equal(tout.cpu().numpy(), want)


@pytest.mark.parametrize("w,h,rng", [(256, 200, 16), (136, 72, 64)])
def test_sharded_motion_search_gives_identical_winners(node, codec, w, h, rng):
    """A frame searched as 1, 2 and 5 stripes (and more stripes than block rows) gives the records of the
    single-device call; with me_local_copy the stripes go through stripe buffers holding exactly what a peer
    would receive (stripe of cur, stripe +- range rows of the reference), so a wrong halo shows up."""
    cur, refp = me_frames(w, h, rng, 0x51, mv=(3, -2))
    mv0, cost0, _ = codec.satd_search(cur, refp, rng, rng)
    tc, tr = D.from_numpy(cur), D.from_numpy(refp)
    nb = (h // 8) * (w // 8)
    origin = tr.data_ptr() + rng * tr.stride(0) + rng
    for local_copy in (0, 1):
        node.set_option("me_local_copy", local_copy)
        for n_stripes in (0, 1, 2, 5, h // 8 + 3):
            best = D.zeros(nb * 2, np.int32)
            D.synchronize()
            node.satd_search(tc.data_ptr(), tc.stride(0), origin, tr.stride(0), w, h, rng, n_stripes, best.data_ptr())
            raw = best.cpu().numpy()
            mv = raw.view(np.int16).reshape(nb, 4)[:, :2]
            cost = raw.view(np.uint32).reshape(nb, 2)[:, 1]
            assert np.array_equal(mv, mv0) and np.array_equal(cost, cost0), (local_copy, n_stripes)
    node.set_option("me_local_copy", 0)


def test_plain_c_host_stream8k():
    """host/stream8k.c: the node API from a C host with no HIP headers, all visible devices, 7680x4320;
    it validates every frame against the single-device calls before timing."""
    exe = os.path.join(ROOT, "host", "stream8k")
    assert os.path.exists(exe), "host/stream8k is not built (make -C host)"
    r = subprocess.run([exe, "0", "50"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["bit_exact_vs_single_device"] is True and line["frames_per_s"] > 100
    # three ranks sharing the device (peer-copy transport): every frame still equals the single-device result
    r = subprocess.run([exe, "3", "20", "1920", "1088"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ranks"] == 3 and line["bit_exact_vs_single_device"] is True


def test_calls_restore_the_callers_device(codec):
    """entry points run on the context's device and put the caller's current device back."""
    import ctypes
    hip, cur = hip_runtime(), ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0 and cur.value == 0
    x = D.zeros(1024, np.int16)
    codec.dct32_fwd_dev(x.data_ptr(), x.clone().data_ptr(), 1)
    D.synchronize()
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0 and cur.value == 0


def test_searches_on_two_streams_do_not_share_scratch(codec, oracle):
    """the search's coefficient scratch is per stream -- two searches of different frames
    enqueued on two streams at once give what they give alone."""
    rng = 8
    frames = [me_frames(512, 256, rng, seed, mv=(1 + seed % 3, -1)) for seed in (11, 12)]
    alone = [codec.satd_search(c, r, rng, rng)[:2] for c, r in frames]
    streams = [codec.stream_create() for _ in frames]
    bufs = []
    for c, r in frames:
        tc, tr = D.from_numpy(c), D.from_numpy(r)
        bufs.append((tc, tr, D.zeros((256 // 8) * (512 // 8) * 2, np.int32)))
    D.synchronize()
    for _ in range(8):                                # interleaved launches, both streams busy at once
        for (tc, tr, best), s in zip(bufs, streams):
            codec.satd_search_dev(tc.data_ptr(), tc.stride(0), tr.data_ptr() + rng * tr.stride(0) + rng, tr.stride(0), 512, 256, rng,
                                  best.data_ptr(), 0, s)
    for s in streams:
        codec.stream_sync(s)
    for (tc, tr, best), (mv0, cost0) in zip(bufs, alone):
        raw = best.cpu().numpy()
        assert np.array_equal(raw.view(np.int16).reshape(-1, 4)[:, :2], mv0)
        assert np.array_equal(raw.view(np.uint32).reshape(-1, 2)[:, 1], cost0)
    for s in streams:
        codec.stream_destroy(s)


# ---- several ranks on the one GPU of the test box ----------------------------------------------------------------
# RCCL refuses two ranks on one device (ncclCommInitAll: invalid usage), and a single-process node then falls back to
# its hipMemcpyPeerAsync transport (SURVEY 8e's alternative).  That makes a real N-rank node on ONE device: every
# rank has its own context, streams, events and slot buffers, the root works zero-copy in the caller's buffers, peers'
# shards and stripes really travel (device-to-device copies instead of ncclSend/ncclRecv) -- the whole schedule and
# partition of the multi-GPU path except the RCCL calls themselves, which test_rccl_executes_on_this_box covers.
@pytest.fixture(scope="module", params=[2, 3, 5])
def multi_node(request):
    n = Node.single_process([0] * request.param)
    assert n.world == request.param and n.local_ranks == list(range(request.param))
    yield n
    n.close()


@pytest.mark.parametrize("w,h,n_frames", [(96, 160, 7), (64, 32, 5), (7680, 4320, 3)])
def test_multi_rank_frame_stream_is_bit_exact(multi_node, oracle, w, h, n_frames):
    """Ragged shards (15 DCT blocks over 2, 3, 5 ranks; 2 blocks over 3 and 5 ranks leaves ranks idle), slots reused,
    two frames in flight; at 7680x4320 every rank transforms its 1/N of the frame and the root's buffers receive the rest."""
    n_d, n_s = (w // 32) * (h // 32), (w // 8) * (h // 8)
    st = multi_node.frame_stream(w, h)
    xin = [(D.from_numpy(_frame(oracle, n_d * 1024, 0x266, f)), D.from_numpy(_frame(oracle, n_s * 64, 0x267, f)))
           for f in range(n_frames)]
    out = [(D.zeros(n_d * 1024, np.int16), D.zeros(n_s, np.int32)) for _ in range(n_frames)]
    D.synchronize()
    tickets = []
    for f in range(n_frames):
        tickets.append(st.push([xin[f][0].data_ptr(), xin[f][1].data_ptr()], [out[f][0].data_ptr(), out[f][1].data_ptr()]))
        if f >= 2:
            st.wait(tickets[f - 2])
            assert np.array_equal(out[f - 2][1].cpu().numpy(), oracle.satd8x8(xin[f - 2][1].cpu().numpy(), threads=32).astype(np.int32)), f - 2
    st.flush()
    for f in range(n_frames):
        assert np.array_equal(out[f][0].cpu().numpy(), oracle.dct32_fwd(xin[f][0].cpu().numpy(), threads=32).ravel()), f
        assert np.array_equal(out[f][1].cpu().numpy(), oracle.satd8x8(xin[f][1].cpu().numpy(), threads=32).astype(np.int32)), f
    st.close()


def test_multi_rank_batch_and_sharded_search(multi_node, codec, oracle):
    n = 10007
    x = oracle.fill_residual(n * 1024, 78)
    tin, tout = D.from_numpy(x), D.zeros(n * 1024, np.int16)
    D.synchronize()
    multi_node.batch_scatter_gather(OP_DCT32_FWD, tin.data_ptr(), tout.data_ptr(), n, 1000)
    assert np.array_equal(tout.cpu().numpy(), oracle.dct32_fwd(x, threads=16).ravel())
    w, h, rng = 200, 136, 24
    cur, refp = me_frames(w, h, rng, 0x52, mv=(-2, 3))
    mv0, cost0, _ = codec.satd_search(cur, refp, rng, rng)
    tc, tr = D.from_numpy(cur), D.from_numpy(refp)
    nb = (h // 8) * (w // 8)
    for n_stripes in (0, multi_node.world + 2):             # one stripe per rank; more stripes than ranks (contiguous runs)
        best = D.zeros(nb * 2, np.int32)
        D.synchronize()
        multi_node.satd_search(tc.data_ptr(), tc.stride(0), tr.data_ptr() + rng * tr.stride(0) + rng, tr.stride(0), w, h, rng, n_stripes, best.data_ptr())
        raw = best.cpu().numpy()
        assert np.array_equal(raw.view(np.int16).reshape(nb, 4)[:, :2], mv0) and np.array_equal(raw.view(np.uint32).reshape(nb, 2)[:, 1], cost0), n_stripes


def test_multi_rank_stream_with_ragged_unit_counts(multi_node, oracle):
    """Unit counts that change from frame to frame -- zero, one, fewer than ranks, not divisible -- through the pipelined
    stream with 2, 3 and 5 ranks: every shard boundary and every skipped (empty) transfer must agree on both ends."""
    caps = [37, 1003, 37]
    st = multi_node.stream([OP_DCT32_INV, OP_SATD8X8, OP_DCT32_FWD], caps)
    z = D.from_numpy(oracle.fill_residual(caps[0] * 1024, 15))
    d = D.from_numpy(oracle.fill_residual(caps[1] * 64, 16))
    zh, dh = z.cpu().numpy(), d.cpu().numpy()
    rs = np.random.RandomState(multi_node.world)
    plans = [[37, 1003, 37], [0, 0, 0], [1, 1, 1], [multi_node.world - 1, multi_node.world + 1, 0], [0, 7, 36]]
    plans += [[int(rs.randint(0, c + 1)) for c in caps] for _ in range(7)]
    outs = []
    for units in plans:                                              # all frames in flight back to back, distinct output buffers
        o = (D.zeros(caps[0] * 1024, np.int16), D.zeros(caps[1], np.int32),
             D.zeros(caps[2] * 1024, np.int16))
        outs.append(o)
        D.synchronize()
        st.push([z.data_ptr(), d.data_ptr(), z.data_ptr()], [t.data_ptr() for t in o], units)
    st.flush()
    for units, (o0, o1, o2) in zip(plans, outs):
        a, b, c = units
        assert np.array_equal(o0.cpu().numpy()[: a * 1024], oracle.dct32_inv(zh[: a * 1024]).ravel()), units
        assert not o0.cpu().numpy()[a * 1024:].any(), units           # nothing written past the frame's units
        assert np.array_equal(o1.cpu().numpy()[:b], oracle.satd8x8(dh[: b * 64]).astype(np.int32)), units
        assert not o1.cpu().numpy()[b:].any(), units
        assert np.array_equal(o2.cpu().numpy()[: c * 1024], oracle.dct32_fwd(zh[: c * 1024]).ravel()), units
        assert not o2.cpu().numpy()[c * 1024:].any(), units
    st.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_rccl_transport_with_n_ranks_under_the_rccl_model(world):
    """The node layer's RCCL code path itself -- the ncclGroupStart/End of ncclSend/ncclRecv it builds per step -- with
    2, 3 and 8 ranks on this one GPU: tests/rccl_model/librccl_model.so (X266HIP_RCCL_LIB) stands in for librccl in a
    separate process and pairs sends with receives the way RCCL does (k-th send a->b with the k-th receive on b from a,
    equal sizes, anything unmatched = the hang it would be).  Frame stream, batch scatter-gather, sharded search and the
    self-test must all equal the single-device results."""
    model = os.path.join(ROOT, "tests", "rccl_model", "librccl_model.so")
    assert os.path.exists(model), "tests/rccl_model/librccl_model.so is not built (make -C tests/rccl_model)"
    env = dict(os.environ, X266HIP_RCCL_LIB=model)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_model", "run_node_under_model.py"), str(world)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ok all" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert "falls back" not in r.stderr                                  # the RCCL transport ran, not the peer-copy fallback


@pytest.mark.parametrize("world", [2, 3, 8])
def test_one_process_per_rank_under_the_rccl_model(world):
    """The node layer as bench.py drives it under torch.distributed.run -- xHipNodeInitRank in `world` PROCESSES, the root
    pushing frames and the peers pushing nothing -- on this one GPU: the model's multi-process mode (shared-memory
    rendezvous, host-staged copies) stands in for librccl.  Frame streams up to 7680x4320, batch scatter-gather and the sharded
    search, all bit-exact against the single-device calls; an unmatched or mis-sized transfer fails instead of hanging."""
    model = os.path.join(ROOT, "tests", "rccl_model", "librccl_model.so")
    assert os.path.exists(model), "tests/rccl_model/librccl_model.so is not built (make -C tests/rccl_model)"
    env = dict(os.environ, X266HIP_RCCL_LIB=model, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_model", "run_ranks_under_model.py"), str(world)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok all" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ok sharded motion search" in r.stdout and "7680x4320" in r.stdout


@pytest.mark.parametrize("n_ranks,size", [(1, (7680, 4320)), (3, (1920, 1088)), (4, (7680, 4320))])
def test_plain_c_host_with_one_process_per_rank(n_ranks, size):
    """host/stream8k_ranks.c: fork before HIP, rank 0 hands the RCCL id to the others through pipes, xHipNodeInitRank in
    every process, the same Push / Flush sequence everywhere -- from plain C.  One rank runs on real RCCL; more than one
    rank on this one GPU needs the RCCL model (multi-process mode)."""
    exe = os.path.join(ROOT, "host", "stream8k_ranks")
    assert os.path.exists(exe), "host/stream8k_ranks is not built (make -C host)"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n_ranks > 1:
        model = os.path.join(ROOT, "tests", "rccl_model", "librccl_model.so")
        assert os.path.exists(model)
        env["X266HIP_RCCL_LIB"] = model
    r = subprocess.run([exe, str(n_ranks), "12", str(size[0]), str(size[1])], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["processes"] == n_ranks and d["bit_exact_vs_single_device"] is True and d["frames"] == 12


# ---- the real thing: two or more devices (BASELINE configs[4]) ---------------------------------------------------------
# Engages on any box that shows >= 2 gfx950 devices and reports as skipped (with the reason) on the one-GPU boxes: single-process
# node over REAL RCCL between distinct HBMs (transport 0 -- no peer-copy fallback, no model), the process-per-rank twin from plain
# C, bench.py --gpus N without torchrun on the command line, and a deliberately slow peer with the input ring overwritten at the
# documented distance.  Everything is compared bit for bit with the single-device calls.
def _device_count():
    return int(x266_amd.load_library().xHipDeviceCount())


@pytest.fixture(scope="module", params=["rccl-between-devices", "one-device-rehearsal"])
def real_node(request):
    """(node, the device of every rank).  "rccl-between-devices" is the tier proper.  "one-device-rehearsal" runs the SAME test
    bodies on every box with three ranks on device 0 over the peer-copy transport, so that the first multi-GPU lease does not
    also have to be the first execution of this file's code (it proves nothing about RCCL, and says so)."""
    if request.param == "one-device-rehearsal":
        devices = [0, 0, 0]
        node = Node.single_process(devices)
        yield node, devices
        node.close()
        return
    n = _device_count()
    if n < 2:
        pytest.skip("multi-device tier: this box shows %d HIP device(s); it engages from 2" % n)
    assert "X266HIP_RCCL_LIB" not in os.environ, "the multi-device tier must talk to the real RCCL"
    devices = list(range(n))
    node = Node.single_process(devices)
    node.set_option("transport", 0)                   # fails if RCCL did not initialise (the library would have fallen back to peer copies)
    ver, path = Node.rccl_info()
    assert ver > 0 and "rccl_model" not in path, (ver, path)
    yield node, devices
    node.close()


def test_multi_device_self_test_and_world(real_node):
    real_node, devices = real_node
    n = len(devices)
    assert real_node.world == n and real_node.local_ranks == list(range(n)) and real_node.drives_root
    if len(set(devices)) == n:                        # (RCCL refuses one device twice in a communicator: the rehearsal has no self-test)
        real_node.self_test()                         # ring send/recv between distinct devices + all-reduce, checked word by word


@pytest.mark.parametrize("w,h,n_frames", [(96, 160, 9), (7680, 4320, 7)])
def test_multi_device_frame_stream_over_rccl(real_node, codec, oracle, w, h, n_frames):
    """The configs[4] stream at its own size over real xGMI: every frame of every lane equals the single-device calls (and the
    first frame the oracle), tickets waited two steps later as documented."""
    real_node, _ = real_node
    n_d, n_s = (w // 32) * (h // 32), (w // 8) * (h // 8)
    st = real_node.frame_stream(w, h)
    xin = [(D.from_numpy(_frame(oracle, n_d * 1024, 0x266, f)), D.from_numpy(_frame(oracle, n_s * 64, 0x267, f))) for f in range(n_frames)]
    want = []
    for a, b in xin:
        c, e = D.empty_like(a), D.empty(n_s, np.int32)
        codec.dct32_fwd_dev(a.data_ptr(), c.data_ptr(), n_d)
        codec.satd8x8_dev(b.data_ptr(), e.data_ptr(), n_s)
        want.append((c, e))
    D.synchronize()
    assert np.array_equal(want[0][0].cpu().numpy(), oracle.dct32_fwd(xin[0][0].cpu().numpy(), threads=32).ravel())
    assert np.array_equal(want[0][1].cpu().numpy(), oracle.satd8x8(xin[0][1].cpu().numpy(), threads=32).astype(np.int32))
    out = [(D.zeros(n_d * 1024, np.int16), D.zeros(n_s, np.int32)) for _ in range(n_frames)]
    D.synchronize()
    tickets = []
    for f in range(n_frames):
        tickets.append(st.push([xin[f][0].data_ptr(), xin[f][1].data_ptr()], [out[f][0].data_ptr(), out[f][1].data_ptr()]))
        if f >= 2:
            st.wait(tickets[f - 2])
            assert D.equal(out[f - 2][0], want[f - 2][0]) and D.equal(out[f - 2][1], want[f - 2][1]), f - 2
    st.flush()
    for f in range(n_frames):
        assert D.equal(out[f][0], want[f][0]) and D.equal(out[f][1], want[f][1]), f
    st.close()


def test_multi_device_batch_scatter_gather_and_sharded_search(real_node, codec, oracle):
    real_node, _ = real_node
    for op, n, chunk in ((OP_DCT32_FWD, 100003, 0), (OP_SATD8X8, 2000001, 0), (OP_DCT32_INV, 4099, 1000)):
        unit = 64 if op == OP_SATD8X8 else 1024
        tin = D.empty(n * unit, np.int16)
        codec.fill_residual_dev(tin.data_ptr(), tin.numel(), 0x300 + op)
        tout = D.zeros(n if op == OP_SATD8X8 else n * 1024, np.int32 if op == OP_SATD8X8 else np.int16)
        ref = D.empty_like(tout)
        {OP_DCT32_FWD: codec.dct32_fwd_dev, OP_DCT32_INV: codec.dct32_inv_dev, OP_SATD8X8: codec.satd8x8_dev}[op](tin.data_ptr(), ref.data_ptr(), n)
        D.synchronize()
        real_node.batch_scatter_gather(op, tin.data_ptr(), tout.data_ptr(), n, chunk)
        assert D.equal(tout, ref), op
    w, h, rng = 1920, 1088, 32
    cur, refp = me_frames(w, h, rng, 0x53, mv=(4, -1))
    tc, tr = D.from_numpy(cur), D.from_numpy(refp)
    nb = (h // 8) * (w // 8)
    origin = tr.data_ptr() + rng * tr.stride(0) + rng
    single = D.zeros(nb * 2, np.int32)
    codec.satd_search_dev(tc.data_ptr(), tc.stride(0), origin, tr.stride(0), w, h, rng, single.data_ptr())
    D.synchronize()
    for n_stripes in (0, real_node.world + 3, 1):
        best = D.zeros(nb * 2, np.int32)
        D.synchronize()
        real_node.satd_search(tc.data_ptr(), tc.stride(0), origin, tr.stride(0), w, h, rng, n_stripes, best.data_ptr())
        assert D.equal(best, single), n_stripes


def test_multi_device_slow_peer_and_input_ring_reuse(real_node, codec):
    """The run-ahead hazard on real hardware: the last rank's device is kept busy with full-frame motion searches (its frame kernels
    lag by milliseconds) while the root pushes 7680x4320 frames from an input ring of X266_STREAM_IN_RING buffers that are REFILLED
    with the next frame's content as early as the header allows (once three later steps have been issued) and an output ring of
    X266_STREAM_OUT_RING.  Every frame must still equal the single-device result."""
    real_node, devices = real_node
    IN_RING, OUT_RING, n_frames = 4, 5, 40
    w, h = 7680, 4320
    n_d, n_s = (w // 32) * (h // 32), (w // 8) * (h // 8)
    peer_index = real_node.world - 1
    peer = real_node.rank_codec(peer_index)
    DP = Dev(peer)                                    # memory on the peer's device, through the peer's context
    sw, sh, srng = 3840, 2160, 64
    pc = DP.random_u8((sh, sw), 1)
    pr = DP.random_u8((sh + 2 * srng, sw + 2 * srng), 2)
    pbest = DP.empty((sh // 8) * (sw // 8) * 2, np.int32)
    pstream = peer.stream_create()
    DP.synchronize()

    def frame_input(f, a, b, stream):
        codec.fill_residual_dev(a.data_ptr(), a.numel(), 0x266, f * 100000007, stream)
        codec.fill_residual_dev(b.data_ptr(), b.numel(), 0x267, f * 100000007, stream)

    fin = [(D.empty(n_d * 1024, np.int16), D.empty(n_s * 64, np.int16)) for _ in range(IN_RING)]
    fout = [(D.zeros(n_d * 1024, np.int16), D.zeros(n_s, np.int32)) for _ in range(OUT_RING)]
    producer = codec.stream_create()
    st = real_node.frame_stream(w, h)
    got = []
    tickets = []
    for f in range(n_frames):
        if f % 4 == 0:                                  # ~2.3 ms of search per call on the peer's device: it stays behind for the whole run
            for _ in range(6):
                peer.satd_search_dev(pc.data_ptr(), pc.stride(0), pr.data_ptr() + srng * pr.stride(0) + srng, pr.stride(0), sw, sh, srng, pbest.data_ptr(), 0, pstream)
        a, b = fin[f % IN_RING]
        frame_input(f, a, b, producer)                  # overwrites frame f - IN_RING's input: exactly three later steps (f-3, f-2, f-1) have been issued
        c, e = fout[f % OUT_RING]
        tickets.append(st.push([a.data_ptr(), b.data_ptr()], [c.data_ptr(), e.data_ptr()], producer_stream=producer))
        if f >= 2:
            st.wait(tickets[f - 2])
            c2, e2 = fout[(f - 2) % OUT_RING]
            got.append((c2.clone(), e2.clone()))
    st.flush()
    for f in (n_frames - 2, n_frames - 1):
        c2, e2 = fout[f % OUT_RING]
        got.append((c2.clone(), e2.clone()))
    D.synchronize()
    peer.stream_sync(pstream)
    st.close()
    a, b = fin[0]
    c1, e1 = D.empty_like(fout[0][0]), D.empty_like(fout[0][1])
    for f in range(n_frames):
        frame_input(f, a, b, 0)
        codec.dct32_fwd_dev(a.data_ptr(), c1.data_ptr(), n_d)
        codec.satd8x8_dev(b.data_ptr(), e1.data_ptr(), n_s)
        D.synchronize()
        assert D.equal(got[f][0], c1) and D.equal(got[f][1], e1), f
    codec.stream_destroy(producer)
    peer.stream_destroy(pstream)


def test_multi_device_process_per_rank_from_plain_c():
    """host/stream8k_ranks: one process per visible device, real RCCL (ncclCommInitRank over the id rank 0 hands out), 7680x4320."""
    n = _device_count()
    if n < 2:
        pytest.skip("multi-device tier: this box shows %d HIP device(s); it engages from 2" % n)
    exe = os.path.join(ROOT, "host", "stream8k_ranks")
    assert os.path.exists(exe), "host/stream8k_ranks is not built (make -C host)"
    env = {k: v for k, v in os.environ.items() if k != "X266HIP_RCCL_LIB"}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([exe, "0", "60", "7680", "4320"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["processes"] == n and d["bit_exact_vs_single_device"] is True and d["frames"] == 60
