"""Host logic for the multi-GPU path: independent blocks shard contiguously
across ranks -- no data-path collective.  One process
per GPU; the only cross-rank traffic is the barrier / max-time reduction and an
optional 8-byte checksum reduction used to validate a sharded run."""
from typing import Tuple


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of rank's units; the first n % world ranks take one extra."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world: %d/%d" % (rank, world))
    if n_units < 0:
        raise ValueError("negative unit count")
    base, extra = divmod(n_units, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def me_stripe(n_block_rows: int, rank: int, world: int, halo_rows: int, total_rows: int):
    """Motion-search partition: horizontal stripes of block rows plus a read-only
    halo of reference rows above and below (sent with the scatter; no exchange)."""
    b, e = shard_range(n_block_rows, rank, world)
    return (b, e), (max(0, b * 8 - halo_rows), min(total_rows, e * 8 + halo_rows))


def combine_checksums(values):
    """Order-independent combination of per-shard sums of uint16 values (mod 2^64)."""
    total = 0
    for v in values:
        total = (total + int(v)) & 0xFFFFFFFFFFFFFFFF
    return total
