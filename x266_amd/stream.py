"""Host logic for BASELINE configs[4]: a stream of frames whose DCT32 and SATD block batches
are sharded across the GPUs of one node -- one process per GPU, torch.distributed (RCCL on
GPUs, gloo in the CPU tests) carrying ONLY the scatter of inputs from the root and the gather
of outputs back.  There is no exchange step between ranks: blocks are
independent (src_tb/dct32.c:75,167-168; satd8x8 is a pure function).

The per-link arithmetic is why the headline scaling figure keeps shards resident instead: one
xGMI link moves ~153 GB/s, i.e. at most 7.5e7 DCT32 blocks/s of input per peer, 18x below what
one GPU transforms; this module is for reporting that end-to-end figure separately
(bench.py --stream8k).
"""
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch

from .shard import shard_range

DCT_BLOCK_SAMPLES = 32 * 32
SATD_BLOCK_SAMPLES = 8 * 8


@dataclass
class FrameGeometry:
    width: int
    height: int

    @property
    def dct_blocks(self) -> int:            # 32x32 luma blocks per frame
        return (self.width // 32) * (self.height // 32)

    @property
    def satd_blocks(self) -> int:           # 8x8 luma blocks per frame
        return (self.width // 8) * (self.height // 8)


def _padded_shard(n_units: int, world: int) -> int:
    return (n_units + world - 1) // world


class ShardedFrameStream:
    """scatter -> per-rank kernels -> gather, for one frame at a time.

    dct_fn(in_tensor, out_tensor, n_blocks) and satd_fn(in_tensor, out_tensor, n_blocks) run the
    rank-local work on tensors living on `device` (the product passes closures over
    Codec.dct32_fwd_dev / satd8x8_dev; the CPU tests pass the oracle)."""

    def __init__(self, geometry: FrameGeometry, device: torch.device, dct_fn: Callable, satd_fn: Callable,
                 dist=None, root: int = 0):
        self.g, self.device, self.dct_fn, self.satd_fn, self.dist, self.root = geometry, device, dct_fn, satd_fn, dist, root
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.dct_per_rank = _padded_shard(geometry.dct_blocks, self.world)
        self.satd_per_rank = _padded_shard(geometry.satd_blocks, self.world)
        self.dct_range = shard_range(geometry.dct_blocks, self.rank, self.world)
        self.satd_range = shard_range(geometry.satd_blocks, self.rank, self.world)
        mk = lambda n, dt: torch.zeros(n, dtype=dt, device=device)
        self.dct_in = mk(self.dct_per_rank * DCT_BLOCK_SAMPLES, torch.int16)
        self.dct_out = mk(self.dct_per_rank * DCT_BLOCK_SAMPLES, torch.int16)
        self.satd_in = mk(self.satd_per_rank * SATD_BLOCK_SAMPLES, torch.int16)
        self.satd_out = mk(self.satd_per_rank, torch.int32)

    # -- root side helpers -------------------------------------------------------------------
    def _split(self, full: Optional[torch.Tensor], n_units: int, unit: int, per_rank: int) -> Optional[List[torch.Tensor]]:
        if self.rank != self.root:
            return None
        parts = []
        for r in range(self.world):
            b, e = shard_range(n_units, r, self.world)
            t = torch.zeros(per_rank * unit, dtype=full.dtype, device=self.device)
            t[: (e - b) * unit] = full[b * unit:e * unit]
            parts.append(t)
        return parts

    def process(self, dct_frame: Optional[torch.Tensor], satd_frame: Optional[torch.Tensor]):
        """dct_frame / satd_frame: on the root, the frame's residual blocks (flat int16); None elsewhere.
        Returns (coefficients, costs) on the root, (None, None) elsewhere."""
        g, d = self.g, self.dist
        if d is None:
            self.dct_in[: g.dct_blocks * DCT_BLOCK_SAMPLES] = dct_frame
            self.satd_in[: g.satd_blocks * SATD_BLOCK_SAMPLES] = satd_frame
        else:
            # int16 is not a collective dtype in RCCL/NCCL (nor gloo): samples travel as bytes
            b8 = lambda parts: None if parts is None else [p.view(torch.uint8) for p in parts]
            d.scatter(self.dct_in.view(torch.uint8), b8(self._split(dct_frame, g.dct_blocks, DCT_BLOCK_SAMPLES, self.dct_per_rank)), src=self.root)
            d.scatter(self.satd_in.view(torch.uint8), b8(self._split(satd_frame, g.satd_blocks, SATD_BLOCK_SAMPLES, self.satd_per_rank)), src=self.root)
        n_d = self.dct_range[1] - self.dct_range[0]
        n_s = self.satd_range[1] - self.satd_range[0]
        self.dct_fn(self.dct_in, self.dct_out, n_d)
        self.satd_fn(self.satd_in, self.satd_out, n_s)
        if d is None:
            return self.dct_out[: g.dct_blocks * DCT_BLOCK_SAMPLES].clone(), self.satd_out[: g.satd_blocks].clone()
        is_root = self.rank == self.root
        dl = [torch.empty_like(self.dct_out) for _ in range(self.world)] if is_root else None
        sl = [torch.empty_like(self.satd_out) for _ in range(self.world)] if is_root else None
        d.gather(self.dct_out.view(torch.uint8), None if dl is None else [t.view(torch.uint8) for t in dl], dst=self.root)
        d.gather(self.satd_out, sl, dst=self.root)
        if not is_root:
            return None, None
        coef = torch.cat([dl[r][: (shard_range(g.dct_blocks, r, self.world)[1] - shard_range(g.dct_blocks, r, self.world)[0]) * DCT_BLOCK_SAMPLES]
                          for r in range(self.world)])
        cost = torch.cat([sl[r][: shard_range(g.satd_blocks, r, self.world)[1] - shard_range(g.satd_blocks, r, self.world)[0]]
                          for r in range(self.world)])
        return coef, cost


class PipelinedFrameStream:
    """The same root-fed frame stream, pipelined: while every rank transforms frame f, frame f+1's
    inputs travel root -> peers and frame f-1's outputs travel peers -> root.

    Transfers are point-to-point (one batch of isend/irecv per stage = one ncclGroupStart/End on
    RCCL, all root links busy at once), not scatter/gather collectives, so the
    ragged shards need no padding and the root sends slices of the frame as they lie.  On a GPU the
    transfers are posted on their own HIP stream and ordered against the kernels with events only;
    two buffer slots per rank (frame parity).  With gloo on CPU tensors (the tests) the same
    schedule runs with blocking waits.

    feed(f) -> (dct_frame, satd_frame) flat int16 tensors on `device`; called on the root only.
    sink(f, coef, cost) is called on the root when frame f is complete; the tensors are views of
    the stream's own slot buffers and are overwritten two frames later.
    """

    SLOTS = 2

    def __init__(self, geometry: FrameGeometry, device: torch.device, dct_fn: Callable, satd_fn: Callable,
                 dist=None, root: int = 0):
        self.g, self.device, self.dct_fn, self.satd_fn, self.dist, self.root = geometry, device, dct_fn, satd_fn, dist, root
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.is_root = self.rank == root
        self.dct_range = shard_range(geometry.dct_blocks, self.rank, self.world)
        self.satd_range = shard_range(geometry.satd_blocks, self.rank, self.world)
        self.n_d = self.dct_range[1] - self.dct_range[0]
        self.n_s = self.satd_range[1] - self.satd_range[0]
        self.cuda = device.type == "cuda"
        mk = lambda n, dt: torch.zeros(max(n, 1), dtype=dt, device=device)
        S = self.SLOTS
        self.dct_in = [mk(self.n_d * DCT_BLOCK_SAMPLES, torch.int16) for _ in range(S)]
        self.dct_out = [mk(self.n_d * DCT_BLOCK_SAMPLES, torch.int16) for _ in range(S)]
        self.satd_in = [mk(self.n_s * SATD_BLOCK_SAMPLES, torch.int16) for _ in range(S)]
        self.satd_out = [mk(self.n_s, torch.int32) for _ in range(S)]
        if self.is_root:                                    # whole-frame result buffers, peers' shards land in place
            self.coef = [mk(geometry.dct_blocks * DCT_BLOCK_SAMPLES, torch.int16) for _ in range(S)]
            self.cost = [mk(geometry.satd_blocks, torch.int32) for _ in range(S)]
        if self.cuda:
            self.comm_stream = torch.cuda.Stream(device=device)
            self.ev_in = [None] * S                         # inputs of the slot have arrived
            self.ev_done = [None] * S                       # kernels of the slot have finished
            self.ev_out = [None] * S                        # outputs of the slot have left / arrived

    # -- helpers ------------------------------------------------------------------------------
    def _ranges(self, n_units, r):
        return shard_range(n_units, r, self.world)

    def _exchange(self, ops):
        if not ops:
            return
        for w in self.dist.batch_isend_irecv(ops):
            w.wait()                                        # GPU: the current (comm) stream waits; CPU: blocks

    def _post_inputs(self, f, feed):
        """root -> peers: frame f's shards (root keeps its own by a local copy)."""
        s = f % self.SLOTS
        d, g = self.dist, self.g
        frames = feed(f) if self.is_root else (None, None)
        ops = []
        P2P = d.P2POp if d is not None else None
        if self.is_root:
            for (full, n_units, unit, mine) in ((frames[0], g.dct_blocks, DCT_BLOCK_SAMPLES, self.dct_in[s]),
                                                (frames[1], g.satd_blocks, SATD_BLOCK_SAMPLES, self.satd_in[s])):
                for r in range(self.world):
                    b, e = self._ranges(n_units, r)
                    if e == b:
                        continue
                    part = full[b * unit:e * unit]
                    if r == self.rank:
                        mine[: (e - b) * unit].copy_(part, non_blocking=True)
                    else:
                        ops.append(P2P(d.isend, part.view(torch.uint8), r))
        else:
            if self.n_d:
                ops.append(P2P(d.irecv, self.dct_in[s][: self.n_d * DCT_BLOCK_SAMPLES].view(torch.uint8), self.root))
            if self.n_s:
                ops.append(P2P(d.irecv, self.satd_in[s][: self.n_s * SATD_BLOCK_SAMPLES].view(torch.uint8), self.root))
        self._exchange(ops)

    def _post_outputs(self, f):
        """peers -> root: frame f's coefficients and costs, straight into the root's frame buffers."""
        s = f % self.SLOTS
        d, g = self.dist, self.g
        ops = []
        P2P = d.P2POp if d is not None else None
        if self.is_root:
            for (full, n_units, unit, mine, n_mine) in ((self.coef[s], g.dct_blocks, DCT_BLOCK_SAMPLES, self.dct_out[s], self.n_d),
                                                        (self.cost[s], g.satd_blocks, 1, self.satd_out[s], self.n_s)):
                for r in range(self.world):
                    b, e = self._ranges(n_units, r)
                    if e == b:
                        continue
                    part = full[b * unit:e * unit]
                    if r == self.rank:
                        part.copy_(mine[: n_mine * unit], non_blocking=True)
                    else:
                        ops.append(P2P(d.irecv, part.view(torch.uint8), r))
        else:
            if self.n_d:
                ops.append(P2P(d.isend, self.dct_out[s][: self.n_d * DCT_BLOCK_SAMPLES].view(torch.uint8), self.root))
            if self.n_s:
                ops.append(P2P(d.isend, self.satd_out[s][: self.n_s].view(torch.uint8), self.root))
        self._exchange(ops)

    def _compute(self, f):
        s = f % self.SLOTS
        self.dct_fn(self.dct_in[s], self.dct_out[s], self.n_d)
        self.satd_fn(self.satd_in[s], self.satd_out[s], self.n_s)

    # -- the schedule -------------------------------------------------------------------------
    def run(self, n_frames: int, feed: Callable, sink: Optional[Callable] = None):
        """Iteration f posts: inputs of frame f, kernels of frame f-1, outputs of frame f-2."""
        S = self.SLOTS
        compute_stream = torch.cuda.current_stream(self.device) if self.cuda else None
        for f in range(n_frames + 2):
            fin, fk, fout = f, f - 1, f - 2
            if fin < n_frames:
                if self.cuda:
                    with torch.cuda.stream(self.comm_stream):
                        if self.ev_done[fin % S] is not None:          # kernels of frame f-2 have read this slot
                            self.comm_stream.wait_event(self.ev_done[fin % S])
                        self.comm_stream.wait_stream(compute_stream)   # whatever produced the fed frame
                        self._post_inputs(fin, feed)
                        self.ev_in[fin % S] = self.comm_stream.record_event()
                else:
                    self._post_inputs(fin, feed)
            if 0 <= fk < n_frames:
                if self.cuda:
                    compute_stream.wait_event(self.ev_in[fk % S])
                    if self.ev_out[fk % S] is not None:                # outputs of frame f-3 have left this slot
                        compute_stream.wait_event(self.ev_out[fk % S])
                    self._compute(fk)
                    self.ev_done[fk % S] = compute_stream.record_event()
                else:
                    self._compute(fk)
            if 0 <= fout < n_frames:
                if self.cuda:
                    with torch.cuda.stream(self.comm_stream):
                        self.comm_stream.wait_event(self.ev_done[fout % S])
                        self._post_outputs(fout)
                        self.ev_out[fout % S] = self.comm_stream.record_event()
                    if self.is_root and sink is not None:
                        self.ev_out[fout % S].synchronize()
                else:
                    self._post_outputs(fout)
                if self.is_root and sink is not None:
                    sink(fout, self.coef[fout % S][: self.g.dct_blocks * DCT_BLOCK_SAMPLES], self.cost[fout % S][: self.g.satd_blocks])
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
