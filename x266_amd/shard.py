"""Partitioning of the multi-GPU path: independent blocks shard contiguously
across ranks, motion search shards into horizontal stripes with a read-only halo -- no data-path
collective, no exchange step.  The arithmetic lives in the C library (xShardRange, xMeStripePlan in
x266_amd/csrc/x266hip_node.cpp: host-only, no device needed) so that the C node layer, bench.py and the
CPU (gloo) tests of the N > 1 logic all partition with the same code.  This module is the binding and
nothing else: without the library it raises, like the rest of the package (a Python restatement exists
only in tests/test_shard_gloo.py, as the cross-check)."""
from typing import Tuple

from .node import me_stripe_plan as _me_stripe_plan, shard_range as _shard_range


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of rank's units; the first n % world ranks take one extra."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world: %d/%d" % (rank, world))
    if n_units < 0:
        raise ValueError("negative unit count")
    return tuple(_shard_range(n_units, rank, world))


def me_stripe(height: int, rng: int, stripe: int, n_stripes: int):
    """Motion-search partition: stripe's block rows [b, e) and the rows [r0, r1) of the padded reference it
    reads (frame coordinates: r0 may be negative, r1 may exceed height, by up to rng)."""
    return _me_stripe_plan(height, rng, stripe, n_stripes)


def combine_checksums(values):
    """Order-independent combination of per-shard sums of uint16 values (mod 2^64)."""
    total = 0
    for v in values:
        total = (total + int(v)) & 0xFFFFFFFFFFFFFFFF
    return total
