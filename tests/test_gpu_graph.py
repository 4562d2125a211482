"""GPU: a frame's worth of small kernels captured once into a HIP graph (xHipGraphBegin / End / Launch)
and replayed over the same buffers with new contents must give exactly what the direct calls give."""
import time

import numpy as np
import pytest

import x266_amd
from _util import me_frames, splitmix64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    return x266_amd.Codec(0)


class FrameJob:
    """1920x1088 luma pipeline on device buffers: planar -> tiles (cur, pred), fused residual+DCT32,
    fused residual+SATD, residual -> forward+inverse DCT32, motion search of cur in a padded reference."""

    def __init__(self, cd, w=1920, h=1088, rng=8):
        self.cd, self.w, self.h, self.rng, self.pad = cd, w, h, rng, 16
        a = cd.alloc
        nt = (w // 16) * (h // 16)
        self.y = [a(w * h), a(w * h)]
        self.u = [a(w * h // 4), a(w * h // 4)]
        self.v = [a(w * h // 4), a(w * h // 4)]
        self.tiles = [a(nt * 512), a(nt * 512)]
        self.coef, self.cost = a(w * h * 2), a(w * h // 64 * 4)
        self.res, self.coef2, self.recon = a(w * h * 2), a(w * h * 2), a(w * h * 2)
        self.refp = a((w + 2 * self.pad) * (h + 2 * self.pad))
        self.best = a((w // 8) * (h // 8) * 8)
        self.n_blk = w * h // 1024                        # intra: one reference set and mode per 32x32 block, the cur plane's bytes as block-major source
        self.irefs, self.imodes, self.icoef = a(self.n_blk * 144), a(self.n_blk), a(w * h * 2)
        self.ctu = a(w * h * 3)                            # round 6: the whole-CTU launch (Y0..Y3 U V) and the chroma costs of the same tiles
        self.cost_c = a(nt * 8)

    def upload(self, seed):
        w, h, pad = self.w, self.h, self.pad
        cur, refp = me_frames(w, h, pad, seed, mv=(2, -1))
        pred = ((splitmix64(seed + 1, 0, w * h) >> np.uint64(9)) & np.uint64(0xFF)).astype(np.uint8).reshape(h, w)
        chroma = ((splitmix64(seed + 2, 0, w * h // 2) >> np.uint64(9)) & np.uint64(0xFF)).astype(np.uint8)
        for i, lum in enumerate((cur, pred)):
            self.y[i].upload(lum)
            self.u[i].upload(chroma[: w * h // 4])
            self.v[i].upload(chroma[w * h // 4:])
        self.refp.upload(refp)
        rs = np.random.RandomState(seed)
        self.irefs.upload(rs.randint(0, 256, self.n_blk * 144).astype(np.uint8))
        self.imodes.upload(rs.randint(0, 35, self.n_blk).astype(np.uint8))

    def enqueue(self, st, with_search=True):
        cd, w, h = self.cd, self.w, self.h
        for i in range(2):
            cd.conv_input_fmt_dev(self.tiles[i].ptr, self.y[i].ptr, self.u[i].ptr, self.v[i].ptr, w, w, h, st)
        cd.dct32_fwd_from_tiles_dev(self.tiles[0].ptr, self.tiles[1].ptr, w, h, self.coef.ptr, st)
        cd.satd8x8_from_tiles_dev(self.tiles[0].ptr, self.tiles[1].ptr, w, h, self.cost.ptr, st)
        cd.residual_luma_dev(self.tiles[0].ptr, self.tiles[1].ptr, w, h, 32, self.res.ptr, st)
        cd.dct32_fwd_inv_dev(self.res.ptr, self.coef2.ptr, self.recon.ptr, w * h // 1024, st)
        cd.intra32_residual_dct32_dev(self.irefs.ptr, self.imodes.ptr, 0, self.y[0].ptr, self.icoef.ptr, self.n_blk, st)
        cd.dct32_fwd_ctu_from_tiles_dev(self.tiles[0].ptr, self.tiles[1].ptr, w, h, self.ctu.ptr, st)
        cd.satd8x8_chroma_from_tiles_dev(self.tiles[0].ptr, self.tiles[1].ptr, w, h, self.cost_c.ptr, self.cost_c.ptr + 4, 2, st)
        if not with_search:
            return
        stride = w + 2 * self.pad
        cd.satd_search_dev(self.y[0].ptr, w, self.refp.ptr + self.pad * stride + self.pad, stride, w, h, self.rng, self.best.ptr, 0, st)

    def results(self):
        w, h = self.w, self.h
        return [self.coef.download(np.int16, w * h), self.cost.download(np.uint32, w * h // 64), self.coef2.download(np.int16, w * h),
                self.recon.download(np.int16, w * h), self.best.download(np.uint32, (w // 8) * (h // 8) * 2), self.icoef.download(np.int16, w * h),
                self.ctu.download(np.int16, w * h * 3 // 2), self.cost_c.download(np.uint32, (w // 16) * (h // 16) * 2)]


def test_graph_replay_equals_direct_calls(codec):
    job = FrameJob(codec)
    st = codec.stream_create()
    try:
        job.upload(10)
        job.enqueue(st)                                    # warm-up outside the capture (sizes the search scratch)
        codec.stream_sync(st)
        codec.graph_begin(st)
        job.enqueue(st)
        graph = codec.graph_end(st)
        for seed in (20, 30):
            job.upload(seed)
            job.enqueue(st)
            codec.stream_sync(st)
            direct = job.results()
            for buf in (job.coef, job.cost, job.coef2, job.recon, job.best, job.icoef, job.ctu, job.cost_c):   # wipe, then replay the graph
                buf.upload(np.zeros(buf.nbytes, np.uint8))
            codec.graph_launch(graph, st)
            codec.stream_sync(st)
            for a, b in zip(direct, job.results()):
                assert np.array_equal(a, b)
            assert direct[0].any() and direct[1].any()
        # the eight small kernels without the search are launch-bound: one submission instead of eight
        codec.graph_begin(st)
        job.enqueue(st, with_search=False)
        small = codec.graph_end(st)
        t = {}
        for name, fn in (("direct", lambda: job.enqueue(st)), ("graph", lambda: codec.graph_launch(graph, st)),
                         ("direct, no search", lambda: job.enqueue(st, with_search=False)),
                         ("graph, no search", lambda: codec.graph_launch(small, st))):
            fn()
            codec.stream_sync(st)
            t0 = time.perf_counter()
            for _ in range(50):
                fn()
            codec.stream_sync(st)
            t[name] = (time.perf_counter() - t0) / 50
        print("frame job: " + ", ".join("%s %.1f us" % (k, v * 1e6) for k, v in t.items()))
        codec.graph_free(small)
        codec.graph_free(graph)
    finally:
        codec.stream_destroy(st)


def test_graph_argument_errors(codec):
    with pytest.raises(x266_amd.X266Error):
        codec.graph_begin(0)                               # the NULL stream cannot be captured
