"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- contiguous shards
per rank, no data-path collective, max-over-ranks timing, checksum reduction.
The per-shard compute is stood in for by the oracle (test infrastructure)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from x266_amd.shard import combine_checksums, me_stripe, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 9, 1000003, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
                assert e0 == b1 and e0 >= b0
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_me_stripes_cover_with_halo():
    rows = 270                                       # 2160 / 8 block rows
    for world in (1, 2, 4, 8):
        covered = 0
        for r in range(world):
            (b, e), (p0, p1) = me_stripe(rows, r, world, 64, 2160)
            covered += e - b
            assert p0 <= max(0, b * 8 - 64) and p1 >= min(2160, e * 8 + 64)
        assert covered == rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_blocks, tmpdir):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from _util import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    b, e = shard_range(n_blocks, rank, world)
    x = orc.fill_residual((e - b) * 1024, 0x266, b * 1024)          # this rank's slice of ONE stream
    z = orc.dct32_fwd(x)
    local = torch.tensor([int(np.uint64(z.view(np.uint16).astype(np.uint64).sum())) & 0x7FFFFFFFFFFFFFFF,
                          e - b], dtype=torch.int64)
    gathered = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.barrier()
    dist.all_gather(gathered, local)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)       # max-over-ranks timing
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(os.path.join(tmpdir, "res.npy"),
                np.array([combine_checksums([int(g[0]) for g in gathered]),
                          sum(int(g[1]) for g in gathered), int(round(float(t[0]) * 10))], dtype=np.int64))
    dist.destroy_process_group()


def test_two_rank_sharded_run_matches_single(oracle, tmp_path):
    n = 301                                                          # odd: ragged shards
    mp.spawn(_worker, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    res = np.load(os.path.join(str(tmp_path), "res.npy"))
    z = oracle.dct32_fwd(oracle.fill_residual(n * 1024, 0x266))
    want = int(z.view(np.uint16).astype(np.uint64).sum())
    assert int(res[0]) == want and int(res[1]) == n and int(res[2]) == 2
