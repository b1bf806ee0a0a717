"""Multi-GPU frame assembly: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests).

The path shards by construction -- pixels x samples are independent once the RNG is keyed
by (seed, pixel, sample) -- so ranks exchange nothing while rendering.  The one exchange
is the final gather of compact tile buffers to rank 0: point-to-point over xGMI, a
gather-to-root uses the root's direct links concurrently (3.1 MB per rank at 1080p / 8
ranks), so no ring and no bucketing is involved.

Partition: rank r owns tiles r, r + world, r + 2*world, ... (interleaved, because per-row
cost is uneven: sky rows cost 1 ray/sample, floor rows 3-5).  Rank buffers are
padded to the same tile count (they differ by at most one tile) so a plain gather works.
"""
import torch
import torch.distributed as dist

from . import abi


def n_tiles(width, height):
    return ((width + abi.TILE - 1) // abi.TILE) * ((height + abi.TILE - 1) // abi.TILE)


def rank_tiles(width, height, rank, world):
    """-> (tile_first, tile_stride, tile_count) of `rank`."""
    total = n_tiles(width, height)
    count = (total - rank + world - 1) // world if total > rank else 0
    return rank, world, count


def padded_count(width, height, world):
    return (n_tiles(width, height) + world - 1) // world


def alloc_tile_buffers(width, height, world, device):
    """(tiles f32, tiles8 u8), each [padded_count, 64, 3], zero-filled."""
    n = max(padded_count(width, height, world), 1)
    return (torch.zeros((n, abi.TILE_PIXELS, 3), dtype=torch.float32, device=device),
            torch.zeros((n, abi.TILE_PIXELS, 3), dtype=torch.uint8, device=device))


def gather_tiles(tiles, tiles8, rank, world, gathered=None, gathered8=None, dst=0, via_cpu=False, force_collective=False):
    """Gather every rank's padded tile buffers on `dst`.  Returns (list, list) on dst,
    (None, None) elsewhere.  world == 1: no collective, the inputs are returned -- unless
    force_collective, which sends a one-rank group through the backend's gather all the same (how the
    tests run RCCL on a one-GPU box).
    via_cpu: stage through host memory (gloo rehearsals with device buffers; RCCL never needs it)."""
    if world == 1 and not force_collective:
        return [tiles], [tiles8]
    if via_cpu:
        host, host8 = gather_tiles(tiles.cpu(), tiles8.cpu() if tiles8 is not None else None, rank, world, dst=dst,
                                   force_collective=force_collective)
        if rank != dst:
            return None, None
        return ([t.to(tiles.device) for t in host],
                [t.to(tiles.device) for t in host8] if host8 is not None else None)
    if _MODE[0] == "all_gather":
        return _all_gather_tiles(tiles, tiles8, rank, world, dst)
    if rank == dst:
        if gathered is None:
            gathered = [torch.empty_like(tiles) for _ in range(world)]
        if gathered8 is None and tiles8 is not None:
            gathered8 = [torch.empty_like(tiles8) for _ in range(world)]
    try:
        dist.gather(tiles, gathered if rank == dst else None, dst=dst)
    except (RuntimeError, NotImplementedError) as exc:
        # a backend without gather-to-root: every rank takes the same branch (the call fails
        # before any communication), so switching collectively is safe
        import sys
        print(f"rt_amd.dist: gather unavailable ({exc}); using all_gather", file=sys.stderr)
        _MODE[0] = "all_gather"
        return _all_gather_tiles(tiles, tiles8, rank, world, dst)
    if tiles8 is not None:
        dist.gather(tiles8, gathered8 if rank == dst else None, dst=dst)
    return (gathered, gathered8) if rank == dst else (None, None)


_MODE = ["gather"]  # "gather" (to the root only) or "all_gather" (fallback: G times the traffic)


def _all_gather_tiles(tiles, tiles8, rank, world, dst):
    parts = [torch.empty_like(tiles) for _ in range(world)]
    dist.all_gather(parts, tiles)
    parts8 = None
    if tiles8 is not None:
        parts8 = [torch.empty_like(tiles8) for _ in range(world)]
        dist.all_gather(parts8, tiles8)
    return (parts, parts8) if rank == dst else (None, None)


def segments(width, height, world):
    """[(rank, tile_first, tile_stride, tile_count)] -- what each gathered buffer holds."""
    return [(r,) + rank_tiles(width, height, r, world) for r in range(world)]
