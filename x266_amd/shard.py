"""Partitioning of the multi-GPU path: independent blocks shard contiguously
across ranks, motion search shards into horizontal stripes with a read-only halo -- no data-path
collective, no exchange step.  The arithmetic lives in the C library (xShardRange, xMeStripePlan in
x266_amd/csrc/x266hip_node.cpp: host-only, no device needed) so that the C node layer, bench.py and the
CPU (gloo) tests of the N > 1 logic all partition with the same code; this module is the binding."""
from typing import Tuple

try:                                           # the C functions when the library can be loaded (it needs libamdhip64 to resolve) ...
    from .node import me_stripe_plan as _me_stripe_plan, shard_range as _shard_range
    from ._lib import load_library as _load
    _load()
    HAVE_C_PLAN = True
except Exception:                              # ... else the same arithmetic in Python (cross-checked in tests/test_shard_gloo.py)
    HAVE_C_PLAN = False


def shard_range_py(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """xShardRange restated: contiguous shards, the first n % world ranks one unit longer."""
    q, r = divmod(n_units, world)
    b = rank * q + min(rank, r)
    return b, b + q + (1 if rank < r else 0)


def me_stripe_py(height: int, rng: int, stripe: int, n_stripes: int):
    """xMeStripePlan restated: block rows dealt like shard_range, reference rows = the stripe's pixel rows +- rng."""
    if height < 8 or height % 8 or rng < 0 or n_stripes < 1 or not (0 <= stripe < n_stripes):
        raise ValueError("bad stripe plan arguments")
    b, e = shard_range_py(height // 8, stripe, n_stripes)
    return (b, e), (b * 8 - rng, e * 8 + rng)


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of rank's units; the first n % world ranks take one extra."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world: %d/%d" % (rank, world))
    if n_units < 0:
        raise ValueError("negative unit count")
    return _shard_range(n_units, rank, world) if HAVE_C_PLAN else shard_range_py(n_units, rank, world)


def me_stripe(height: int, rng: int, stripe: int, n_stripes: int):
    """Motion-search partition: stripe's block rows [b, e) and the rows [r0, r1) of the padded reference it
    reads (frame coordinates: r0 may be negative, r1 may exceed height, by up to rng)."""
    return _me_stripe_plan(height, rng, stripe, n_stripes) if HAVE_C_PLAN else me_stripe_py(height, rng, stripe, n_stripes)


def combine_checksums(values):
    """Order-independent combination of per-shard sums of uint16 values (mod 2^64)."""
    total = 0
    for v in values:
        total = (total + int(v)) & 0xFFFFFFFFFFFFFFFF
    return total
