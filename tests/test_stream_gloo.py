"""CPU, world_size 2 over gloo: BASELINE configs[4]'s host path -- scatter a frame's DCT32 and SATD
block batches from the root, transform each shard, gather the results -- must give exactly what one
process computes.  The per-shard compute is stood in for by the oracle (test infrastructure)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from x266_amd.stream import FrameGeometry, PipelinedFrameStream, ShardedFrameStream


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_fns():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _util import Oracle
    orc = Oracle()

    def dct_fn(tin, tout, n):
        if n:
            tout[: n * 1024] = torch.from_numpy(orc.dct32_fwd(tin[: n * 1024].numpy()).ravel())

    def satd_fn(tin, tout, n):
        if n:
            tout[:n] = torch.from_numpy(orc.satd8x8(tin[: n * 64].numpy()).astype(np.int32))

    return orc, dct_fn, satd_fn


def _worker(rank, world, port, w, h, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, dct_fn, satd_fn = _oracle_fns()
    geo = FrameGeometry(w, h)
    st = ShardedFrameStream(geo, torch.device("cpu"), dct_fn, satd_fn, dist=dist)
    for frame in range(2):
        fd = fs = None
        if rank == 0:
            fd = torch.from_numpy(orc.fill_residual(geo.dct_blocks * 1024, 0x266, frame * 10 ** 7))
            fs = torch.from_numpy(orc.fill_residual(geo.satd_blocks * 64, 0x267, frame * 10 ** 7))
        coef, cost = st.process(fd, fs)
        if rank == 0:
            np.save(os.path.join(tmpdir, "coef%d.npy" % frame), coef.numpy())
            np.save(os.path.join(tmpdir, "cost%d.npy" % frame), cost.numpy())
        else:
            assert coef is None and cost is None
    dist.destroy_process_group()


def _pipe_worker(rank, world, port, w, h, n_frames, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, dct_fn, satd_fn = _oracle_fns()
    geo = FrameGeometry(w, h)
    st = PipelinedFrameStream(geo, torch.device("cpu"), dct_fn, satd_fn, dist=dist)

    def feed(f):
        return (torch.from_numpy(orc.fill_residual(geo.dct_blocks * 1024, 0x266, f * 10 ** 7)),
                torch.from_numpy(orc.fill_residual(geo.satd_blocks * 64, 0x267, f * 10 ** 7)))

    seen = []

    def sink(f, coef, cost):
        seen.append(f)
        np.save(os.path.join(tmpdir, "pcoef%d.npy" % f), coef.numpy())
        np.save(os.path.join(tmpdir, "pcost%d.npy" % f), cost.numpy())

    st.run(n_frames, feed, sink)
    assert seen == (list(range(n_frames)) if rank == 0 else [])
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 96, 160), (3, 64, 32)])
def test_pipelined_stream_equals_single_process(oracle, tmp_path, world, w, h):
    """Point-to-point, double-buffered schedule: 5 frames (slots are reused twice), ragged shards;
    3 ranks over a 2-block frame leaves one rank without DCT work."""
    n_frames = 5
    mp.spawn(_pipe_worker, args=(world, _free_port(), w, h, n_frames, str(tmp_path)), nprocs=world, join=True)
    geo = FrameGeometry(w, h)
    for f in range(n_frames):
        x = oracle.fill_residual(geo.dct_blocks * 1024, 0x266, f * 10 ** 7)
        d = oracle.fill_residual(geo.satd_blocks * 64, 0x267, f * 10 ** 7)
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "pcoef%d.npy" % f)), oracle.dct32_fwd(x).ravel())
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "pcost%d.npy" % f)), oracle.satd8x8(d).astype(np.int32))


def test_pipelined_single_process(oracle):
    orc, dct_fn, satd_fn = _oracle_fns()
    geo = FrameGeometry(64, 96)
    st = PipelinedFrameStream(geo, torch.device("cpu"), dct_fn, satd_fn, dist=None)
    got = {}
    st.run(3, lambda f: (torch.from_numpy(oracle.fill_residual(geo.dct_blocks * 1024, 5, f)),
                         torch.from_numpy(oracle.fill_residual(geo.satd_blocks * 64, 6, f))),
           lambda f, c, s: got.__setitem__(f, (c.numpy().copy(), s.numpy().copy())))
    for f in range(3):
        assert np.array_equal(got[f][0], oracle.dct32_fwd(oracle.fill_residual(geo.dct_blocks * 1024, 5, f)).ravel())
        assert np.array_equal(got[f][1], oracle.satd8x8(oracle.fill_residual(geo.satd_blocks * 64, 6, f)).astype(np.int32))


def test_two_rank_stream_equals_single_process(oracle, tmp_path):
    w, h = 96, 160                                       # 15 DCT blocks, 240 SATD blocks: ragged over 2 ranks
    mp.spawn(_worker, args=(2, _free_port(), w, h, str(tmp_path)), nprocs=2, join=True)
    geo = FrameGeometry(w, h)
    for frame in range(2):
        x = oracle.fill_residual(geo.dct_blocks * 1024, 0x266, frame * 10 ** 7)
        d = oracle.fill_residual(geo.satd_blocks * 64, 0x267, frame * 10 ** 7)
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "coef%d.npy" % frame)), oracle.dct32_fwd(x).ravel())
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "cost%d.npy" % frame)), oracle.satd8x8(d).astype(np.int32))


def test_single_process_path(oracle):
    orc, dct_fn, satd_fn = _oracle_fns()
    geo = FrameGeometry(64, 64)
    st = ShardedFrameStream(geo, torch.device("cpu"), dct_fn, satd_fn, dist=None)
    x = oracle.fill_residual(geo.dct_blocks * 1024, 1)
    d = oracle.fill_residual(geo.satd_blocks * 64, 2)
    coef, cost = st.process(torch.from_numpy(x), torch.from_numpy(d))
    assert np.array_equal(coef.numpy(), oracle.dct32_fwd(x).ravel())
    assert np.array_equal(cost.numpy(), oracle.satd8x8(d).astype(np.int32))
    assert (FrameGeometry(7680, 4320).dct_blocks, FrameGeometry(7680, 4320).satd_blocks) == (32400, 518400)
