"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- contiguous shards
per rank, no data-path collective, max-over-ranks timing, checksum reduction.
The per-shard compute is stood in for by the oracle (test infrastructure)."""
import os
import socket
import sys

import importlib

import numpy as np
import pytest

from x266_amd.shard import combine_checksums, me_stripe, shard_range


class _Lazy:
    """torch is imported where it is USED -- in the worker PROCESSES of these tests -- never by pytest's own process: a `-m gpu` session
    collects every file, and torch's bundled HIP runtime / RCCL (same SONAMEs as ROCm's) would then serve the whole GPU test process
    (tests/conftest.py, tests/test_gpu_runtime.py); and a CPU session that loaded libx266hip.so first (ROCm's runtime) and torch
    afterwards ended in `double free or corruption` at interpreter exit (two owners of one runtime's state)."""

    def __init__(self, name):
        self._name = name

    def __getattr__(self, attr):
        if attr.startswith("__"):                                       # pytest's collector probes module globals (__test__, ...): not a use
            raise AttributeError(attr)
        return getattr(importlib.import_module(self._name), attr)


torch, dist = _Lazy("torch"), _Lazy("torch.distributed")


def _spawn(fn, args, nprocs):
    """torch.multiprocessing.spawn without torch in the parent: `nprocs` fresh interpreters (spawn context) run fn(rank, *args)"""
    import multiprocessing
    ctx = multiprocessing.get_context("spawn")
    procs = [ctx.Process(target=fn, args=(r,) + tuple(args)) for r in range(nprocs)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    codes = [p.exitcode for p in procs]
    assert codes == [0] * nprocs, "worker exit codes %r" % (codes,)


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
    height, rng = 2160, 64                           # 270 block rows
    for n_stripes in (1, 2, 4, 5, 8, 270, 300):
        nxt = 0
        for s in range(n_stripes):
            (b, e), (r0, r1) = me_stripe(height, rng, s, n_stripes)
            assert b == nxt and e >= b               # contiguous, in order
            nxt = e
            assert (r0, r1) == (b * 8 - rng, e * 8 + rng)      # every row a candidate of the stripe's blocks can touch
        assert nxt == height // 8
    with pytest.raises(ValueError):
        me_stripe(2161, 64, 0, 2)


def _me_worker(rank, world, port, w, h, rng, n_stripes, tmpdir):
    """Sharded motion search over gloo: the root sends every stripe of `cur` and the stripe's halo rows of the
    padded reference to its owner, owners search ONLY what they received (so a short halo shows up as a wrong
    winner), records return to the root.  Partition = the C library's plan, compute = the oracle."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from _util import Oracle, me_frames
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    cur = refp = None
    if rank == 0:
        cur, refp = me_frames(w, h, rng, 0x3E, mv=(4, -3))
    mv_all = np.zeros(((h // 8) * (w // 8), 2), np.int16)
    cost_all = np.zeros((h // 8) * (w // 8), np.uint32)
    bw = w // 8
    for s in range(n_stripes):
        owner = [r for r in range(world) if shard_range(n_stripes, r, world)[0] <= s < shard_range(n_stripes, r, world)[1]][0]
        (b, e), (r0, r1) = me_stripe(h, rng, s, n_stripes)
        if e == b:
            continue
        if rank == 0:
            c = torch.from_numpy(cur[b * 8:e * 8].copy())
            r = torch.from_numpy(refp[r0 + rng:r1 + rng].copy())          # padded array row = frame row + rng
            if owner != 0:
                dist.send(c, owner)
                dist.send(r, owner)
        elif rank == owner:
            c = torch.empty(((e - b) * 8, w), dtype=torch.uint8)
            r = torch.empty((r1 - r0, w + 2 * rng), dtype=torch.uint8)
            dist.recv(c, 0)
            dist.recv(r, 0)
        if rank == owner:
            mv, cost, _ = orc.satd_search(c.numpy(), r.numpy(), rng, rng)
            out = torch.from_numpy(np.concatenate([mv.astype(np.int32).ravel(), cost.astype(np.int64).astype(np.int32)]))
            if owner != 0:
                dist.send(out, 0)
        if rank == 0:
            if owner != 0:
                out = torch.empty((e - b) * bw * 3, dtype=torch.int32)
                dist.recv(out, owner)
            o = out.numpy()
            n = (e - b) * bw
            mv_all[b * bw:e * bw] = o[: 2 * n].reshape(n, 2)
            cost_all[b * bw:e * bw] = o[2 * n:].astype(np.uint32)
    if rank == 0:
        np.save(os.path.join(tmpdir, "me_mv.npy"), mv_all)
        np.save(os.path.join(tmpdir, "me_cost.npy"), cost_all)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_stripes", [(2, 2), (2, 5)])
def test_two_rank_sharded_motion_search_matches_whole_frame(oracle, tmp_path, world, n_stripes):
    from _util import me_frames
    w, h, rng = 48, 40, 6                             # 5 block rows: ragged stripes
    _spawn(_me_worker, (world, _free_port(), w, h, rng, n_stripes, str(tmp_path)), world)
    cur, refp = me_frames(w, h, rng, 0x3E, mv=(4, -3))
    mv, cost, _ = oracle.satd_search(cur, refp, rng, rng)
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "me_mv.npy")), mv)
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "me_cost.npy")), cost)


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
    _spawn(_worker, (2, _free_port(), n, str(tmp_path)), 2)
    res = np.load(os.path.join(str(tmp_path), "res.npy"))
    z = oracle.dct32_fwd(oracle.fill_residual(n * 1024, 0x266))
    want = int(z.view(np.uint16).astype(np.uint64).sum())
    assert int(res[0]) == want and int(res[1]) == n and int(res[2]) == 2


def _shard_range_py(n_units, rank, world):
    """xShardRange restated (the cross-check; the package itself has only the C functions): contiguous shards, the first n % world ranks one unit longer."""
    q, r = divmod(n_units, world)
    b = rank * q + min(rank, r)
    return b, b + q + (1 if rank < r else 0)


def _me_stripe_py(height, rng, stripe, n_stripes):
    """xMeStripePlan restated: block rows dealt like shard_range, reference rows = the stripe's pixel rows +- rng."""
    b, e = _shard_range_py(height // 8, stripe, n_stripes)
    return (b, e), (b * 8 - rng, e * 8 + rng)


def test_the_c_plan_functions_equal_a_python_restatement():
    """xShardRange / xMeStripePlan (x266hip_node.cpp), which the node layer, bench.py and x266_amd.shard all use, against the arithmetic
    written out here; and x266_amd.shard is the C binding (no second implementation in the package)."""
    from x266_amd import shard
    from x266_amd.node import me_stripe_plan
    assert not hasattr(shard, "shard_range_py") and not hasattr(shard, "me_stripe_py")
    for n in (0, 1, 7, 8, 9, 1000, 32400, 518400, (1 << 20) + 3):
        for world in (1, 2, 3, 5, 8):
            for rank in range(world):
                assert _shard_range_py(n, rank, world) == tuple(shard_range(n, rank, world))
    for height in (8, 64, 544, 1080 - 1080 % 8, 2160):
        for rng in (0, 1, 16, 64):
            for n_stripes in (1, 2, 3, 8, 300):
                for stripe in range(min(n_stripes, 9)):
                    assert _me_stripe_py(height, rng, stripe, n_stripes) == me_stripe_plan(height, rng, stripe, n_stripes) == me_stripe(height, rng, stripe, n_stripes)


def test_bench_spawns_its_own_ranks_when_run_plainly():
    """`python bench.py --gpus 2` without WORLD_SIZE must become the torch.distributed.run job itself instead of
    refusing.  Without a GPU both ranks stop at "needs an MI355X" -- which proves that two ranks were started and each got as far as
    the device check (a GPU box runs the same path to the end in tests/test_gpu_bench.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import x266_amd
    if x266_amd.load_library().xHipDeviceCount() > 0:
        pytest.skip("a GPU box runs the real thing (tests/test_gpu_bench.py)")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-also"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    # at least one rank got as far as the device check; the launcher may tear the second one down the moment the first has
    # failed (1 run in ~20 here), so "both messages" would be a race -- that the job WAS a torch.distributed.run of two ranks shows
    # in the launcher's own failure report
    assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]
    assert "ChildFailedError" in r.stderr or "torch.distributed" in r.stderr, r.stderr[-2000:]
    assert "needs torch.distributed.run" not in r.stderr
