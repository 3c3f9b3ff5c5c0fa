"""Sharding of a many-chunk buffer over the GPUs of one node.

Chunks are independent (the only intra-chunk couplings are the bstarts table and the contiguous
packing, blosc/blosc.c:816, :1845-1856), so the path needs NO data-path collective: rank r owns a
contiguous range of chunks and runs the single-GPU batched call on it.  The only exchange is the
per-chunk `cbytes` table (4 bytes per chunk) so that every rank knows the global layout; it is a
`torch.distributed.all_gather` — RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.
"""
from typing import List, Tuple


def chunk_range(nchunks: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of chunk indices owned by `rank` (chunk c -> GPU floor(c*G/nchunks))."""
    # smallest c with floor(c*world/nchunks) >= rank
    lo = -(-rank * nchunks // world)
    hi = -(-(rank + 1) * nchunks // world)
    return lo, min(hi, nchunks)


def owner_of(chunk: int, nchunks: int, world: int) -> int:
    return chunk * world // nchunks


def gather_cbytes(local_cbytes: List[int], nchunks: int, device=None):
    """All ranks contribute the cbytes of their chunks; returns (global cbytes list, exclusive offsets).

    Uses all_gather on equal-sized padded tensors (ranks may own one chunk more or less)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    rank = dist.get_rank()
    per = -(-nchunks // world)
    buf = torch.full((per,), -1, dtype=torch.int32, device=device)
    buf[: len(local_cbytes)] = torch.tensor(local_cbytes, dtype=torch.int32, device=device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    table = []
    for r in range(world):
        lo, hi = chunk_range(nchunks, world, r)
        table.extend(int(v) for v in out[r][: hi - lo].tolist())
    offsets, acc = [], 0
    for c in table:
        offsets.append(acc)
        acc += max(c, 0)
    return table, offsets
