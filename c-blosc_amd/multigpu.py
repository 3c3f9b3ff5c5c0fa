"""Sharding of a many-chunk buffer over the GPUs of one node.

Chunks are independent (the only intra-chunk couplings are the bstarts table and the contiguous
packing, blosc/blosc.c:816, :1845-1856), so the path needs NO data-path collective: rank r owns a
contiguous range of chunks and runs the single-GPU batched call on it.  Two exchanges follow it, both
optional for the data path: the per-chunk `cbytes` table (4 bytes per chunk) so that every rank knows the
global layout - a `torch.distributed.all_gather`, RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU
tests - and, for callers that want one container, the compressed payloads themselves (gather_payload /
scatter_payload below: an all-gather-v with the counts the table gives).
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


# ---------------------------------------------------------------------------------------------
# Consolidation of the compressed payloads.  After the batched call every rank holds
# the chunks of its own range; a container (a file, a message) wants them back to back in chunk order.
# The exchange is an all-gather-v: rank r contributes sum(cbytes[lo_r:hi_r]) bytes, known to everybody
# from gather_cbytes().  It is written as ONE group of point-to-point operations
# (torch.distributed.batch_isend_irecv = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on RCCL: each
# pair of GPUs uses its own xGMI link, ~153 GB/s; nothing is padded to the largest contribution the
# way an all_gather of equal-sized tensors would) and runs unchanged on gloo in the CPU tests.
# The chunks never need re-encoding: their bstarts are offsets from the chunk's own start
# (blosc/blosc.c:816, :1845-1856), so a chunk can be moved as one opaque run of cbytes bytes.
# ---------------------------------------------------------------------------------------------
def rank_byte_ranges(table: List[int], nchunks: int, world: int):
    """[(byte offset, byte count)] of every rank's contribution inside the consolidated container"""
    out, acc = [], 0
    for r in range(world):
        lo, hi = chunk_range(nchunks, world, r)
        n = sum(max(c, 0) for c in table[lo:hi])
        out.append((acc, n))
        acc += n
    return out


def _default_device():
    """Where a rank WITHOUT chunks allocates its (empty) tensors: the current GPU under nccl / RCCL (a CPU tensor cannot be posted
    there), the CPU under gloo."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def pack_local(comp_rows, local_cbytes: List[int], device=None):
    """This rank's chunks back to back, in chunk order: comp_rows[i][:cbytes[i]] concatenated (device-side copies).
    `device`: where to allocate when the rank owns no chunk (default: see _default_device)."""
    import torch
    total = sum(max(c, 0) for c in local_cbytes)
    first = comp_rows[0] if len(local_cbytes) else None
    out = torch.empty((total,), dtype=torch.uint8, device=first.device if first is not None else (device if device is not None else _default_device()))
    acc = 0
    for row, c in zip(comp_rows, local_cbytes):
        if c > 0:
            out[acc:acc + c].copy_(row[:c])
            acc += c
    return out


def gather_payload(packed_local, table: List[int], nchunks: int, dst=None):
    """All-gather-v of the packed compressed bytes.  dst=None: every rank gets the whole container;
    dst=r: only rank r does (the others return None).  Returns (container tensor, per-chunk byte offsets)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    ranges = rank_byte_ranges(table, nchunks, world)
    total = ranges[-1][0] + ranges[-1][1]
    assert packed_local.numel() == ranges[rank][1], (packed_local.numel(), ranges[rank])
    offsets, acc = [], 0
    for c in table:
        offsets.append(acc)
        acc += max(c, 0)
    receivers = list(range(world)) if dst is None else [dst]
    container = torch.empty((total,), dtype=torch.uint8, device=packed_local.device) if rank in receivers else None
    ops = []
    if container is not None:
        o, n = ranges[rank]
        container[o:o + n].copy_(packed_local)
        for r in range(world):
            if r != rank and ranges[r][1]:
                ops.append(dist.P2POp(dist.irecv, container[ranges[r][0]:ranges[r][0] + ranges[r][1]], r))
    if packed_local.numel():
        for r in receivers:
            if r != rank:
                ops.append(dist.P2POp(dist.isend, packed_local, r))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return container, offsets


def scatter_payload(container, table: List[int], nchunks: int, src: int = 0, device=None):
    """The inverse for decompression: rank `src` holds the consolidated container; every rank receives the bytes of
    its own chunk range (returned as one packed tensor + the byte offset of each of its chunks inside it)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    ranges = rank_byte_ranges(table, nchunks, world)
    lo, hi = chunk_range(nchunks, world, rank)
    mine = torch.empty((ranges[rank][1],), dtype=torch.uint8,
                       device=container.device if container is not None else (device if device is not None else _default_device()))
    ops = []
    if rank == src:
        o, n = ranges[rank]
        mine.copy_(container[o:o + n])
        for r in range(world):
            if r != rank and ranges[r][1]:
                ops.append(dist.P2POp(dist.isend, container[ranges[r][0]:ranges[r][0] + ranges[r][1]], r))
    elif mine.numel():
        ops.append(dist.P2POp(dist.irecv, mine, src))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    local_off, acc = [], 0
    for c in table[lo:hi]:
        local_off.append(acc)
        acc += max(c, 0)
    return mine, local_off
