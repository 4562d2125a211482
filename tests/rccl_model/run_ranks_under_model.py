#!/usr/bin/env python3
"""ONE PROCESS PER RANK on one GPU, with X266HIP_RCCL_LIB = the RCCL model in its multi-process mode: the node layer
exactly as bench.py drives it under torch.distributed.run (xHipNodeInitRank, the root pushes frames, the peers push
nothing), which real RCCL cannot run on a one-GPU box.  The parent starts `world` children of itself; rank 0 creates
the id and checks every result against the single-device calls of the same library.

    run_ranks_under_model.py <world>            parent
    run_ranks_under_model.py <world> <rank> <id-file>   child
"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(world, rank, id_file):
    import x266_amd
    from x266_amd.node import Node, OP_DCT32_FWD
    if rank == 0:
        uid = Node.unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            assert time.time() - t0 < 60, "no id from rank 0"
            time.sleep(0.01)
        uid = open(id_file, "rb").read()
    codec = x266_amd.Codec(0)
    node = Node.for_rank(0, rank, world, uid)
    assert node.world == world and node.drives_root == (rank == 0)
    node.self_test()
    root = rank == 0

    def dev(arr):
        b = codec.alloc(max(arr.nbytes, 16))
        b.upload(arr)
        return b

    rs = np.random.RandomState(11)                                   # same stream on every rank: only the root uses the data
    for (w, h, n_frames) in ((96, 160, 6), (64, 32, 4), (1920, 1088, 3), (7680, 4320, 3)):
        nd, ns = (w // 32) * (h // 32), (w // 8) * (h // 8)
        st = node.frame_stream(w, h)
        frames = [(rs.randint(-255, 256, nd * 1024).astype(np.int16), rs.randint(-255, 256, ns * 64).astype(np.int16)) for _ in range(n_frames)]
        bufs = [(dev(a), dev(b), codec.alloc(nd * 2048), codec.alloc(ns * 4)) for a, b in frames] if root else [None] * n_frames
        for b in bufs:
            if root:
                st.push([b[0].ptr, b[1].ptr], [b[2].ptr, b[3].ptr])
            else:
                st.push()
        st.flush()
        if root:
            for (a, b), (da, db, dc, de) in zip(frames, bufs):
                assert np.array_equal(dc.download(np.int16, nd * 1024), codec.dct32_fwd(a).ravel()), (w, h)
                assert np.array_equal(de.download(np.uint32, ns), codec.satd8x8(b)), (w, h)
            print("ok frame stream %dx%d, %d frames, %d processes" % (w, h, n_frames, world), flush=True)
        st.close()
        del bufs

    # unit counts that change from frame to frame: every process passes the same counts, only the root has buffers
    from x266_amd.node import OP_DCT32_INV, OP_SATD8X8
    caps = [19, 301]
    st = node.stream([OP_DCT32_INV, OP_SATD8X8], caps)
    zin = rs.randint(-255, 256, caps[0] * 1024).astype(np.int16)
    din = rs.randint(-255, 256, caps[1] * 64).astype(np.int16)
    plans = [[19, 301], [0, 0], [1, 1], [world - 1, world + 1], [0, 5], [7, 0], [18, 300]]
    if root:
        dz, dd = dev(zin), dev(din)
        outs = [(codec.alloc(caps[0] * 2048), codec.alloc(caps[1] * 4)) for _ in plans]
    for k, units in enumerate(plans):
        if root:
            st.push([dz.ptr, dd.ptr], [outs[k][0].ptr, outs[k][1].ptr], units)
        else:
            st.push(None, None, units)
    st.flush()
    if root:
        for units, (o0, o1) in zip(plans, outs):
            a, b = units
            assert np.array_equal(o0.download(np.int16, caps[0] * 1024)[: a * 1024], codec.dct32_inv(zin[: a * 1024]).ravel()), units
            assert np.array_equal(o1.download(np.uint32, caps[1])[:b], codec.satd8x8(din[: b * 64])), units
        print("ok ragged unit counts", flush=True)
    # an argument error EVERY rank sees identically and that posts nothing (more units than the stream was created for) is refused on
    # every rank and costs nothing: the stream and the node go on (the root-only refusals -- NULL / unaligned / output in
    # flight -- do cost the node with one process per GPU, the peers having posted their side)
    too_many = [caps[0] + 1, caps[1]]
    try:
        if root:
            st.push([dz.ptr, dd.ptr], [outs[0][0].ptr, outs[0][1].ptr], too_many)
        else:
            st.push(None, None, too_many)
        raise AssertionError("units beyond max_units were accepted")
    except x266_amd.X266Error as e:
        assert "exceed" in str(e), e
    if root:
        st.push([dz.ptr, dd.ptr], [outs[1][0].ptr, outs[1][1].ptr], [3, 9])
    else:
        st.push(None, None, [3, 9])
    st.flush()
    if root:
        assert np.array_equal(outs[1][0].download(np.int16, 3 * 1024), codec.dct32_inv(zin[: 3 * 1024]).ravel())
        assert np.array_equal(outs[1][1].download(np.uint32, 9), codec.satd8x8(din[: 9 * 64]))
        print("ok a rank-symmetric argument error leaves the node usable", flush=True)
    st.close()

    n = 5003
    x = rs.randint(-255, 256, n * 1024).astype(np.int16)
    if root:
        di, do = dev(x), codec.alloc(n * 2048)
    node.batch_scatter_gather(OP_DCT32_FWD, di.ptr if root else 0, do.ptr if root else 0, n, 700)
    if root:
        assert np.array_equal(do.download(np.int16, n * 1024), codec.dct32_fwd(x).ravel())
        print("ok batch scatter-gather", flush=True)

    w, h, rng = 200, 136, 24
    cur = rs.randint(0, 256, (h, w)).astype(np.uint8)
    refp = rs.randint(0, 256, (h + 2 * rng, w + 2 * rng)).astype(np.uint8)
    nb = (h // 8) * (w // 8)
    if root:
        mv0, cost0, _ = codec.satd_search(cur, refp, rng, rng)
        dc, dr = dev(cur), dev(refp)
    for n_stripes in (0, world + 2):
        if root:
            db = codec.alloc(nb * 8)
            node.satd_search(dc.ptr, w, dr.ptr + rng * (w + 2 * rng) + rng, w + 2 * rng, w, h, rng, n_stripes, db.ptr)
            raw = db.download(np.uint8, nb * 8)
            assert np.array_equal(raw.view(np.int16).reshape(nb, 4)[:, :2], mv0) and np.array_equal(raw.view(np.uint32).reshape(nb, 2)[:, 1], cost0)
        else:
            node.satd_search(0, w, 0, w + 2 * rng, w, h, rng, n_stripes, 0)
    if root:
        print("ok sharded motion search", flush=True)
    node.close()
    print("ok rank %d" % rank, flush=True)


def parent(world):
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        id_file = os.path.join(d, "id")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(world), str(r), id_file]) for r in range(world)]
        rcs = []
        for p in procs:
            try:
                rcs.append(p.wait(timeout=600))
            except subprocess.TimeoutExpired:
                p.kill()
                rcs.append(-9)
    if any(rcs):
        print("exit codes", rcs)
        sys.exit(1)
    print("ok all")


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        child(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
    else:
        parent(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
