"""CPU, world_size 2 over gloo: the chunk partition and the cbytes exchange used for N > 1 GPUs.
(The compression itself is done with the oracle here — there is no GPU on this box; what is under
test is the sharding logic, which is identical on RCCL.)"""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _multigpu():
    spec = importlib.util.spec_from_file_location("bamd_multigpu", os.path.join(ROOT, "c-blosc_amd", "multigpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_partition_is_a_contiguous_cover():
    m = _multigpu()
    for nchunks in [1, 2, 7, 8, 128, 129, 4096]:
        for world in [1, 2, 3, 4, 8]:
            seen = []
            for r in range(world):
                lo, hi = m.chunk_range(nchunks, world, r)
                assert 0 <= lo <= hi <= nchunks
                seen.extend(range(lo, hi))
                for c in range(lo, hi):
                    assert m.owner_of(c, nchunks, world) == r
            assert seen == list(range(nchunks))


def _worker(rank, world, port, nchunks, q):
    import ctypes as C
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import DATASETS, orc_compress
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _multigpu()
    O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    O.orc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int]
    lo, hi = m.chunk_range(nchunks, world, rank)
    local = []
    for c in range(lo, hi):
        data = DATASETS["bench19" if c % 2 == 0 else "randwalk"](20000 + 1000 * c)
        r, _ = orc_compress(O, data, 8, 5, 1, "lz4")
        local.append(r)
    table, offsets = m.gather_cbytes(local, nchunks)
    dist.barrier()
    q.put((rank, table, offsets))
    dist.destroy_process_group()


def test_cbytes_exchange_world2(oracle):
    import torch.multiprocessing as mp
    from helpers import DATASETS, orc_compress
    nchunks, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, nchunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = []
    for c in range(nchunks):
        data = DATASETS["bench19" if c % 2 == 0 else "randwalk"](20000 + 1000 * c)
        want.append(orc_compress(oracle, data, 8, 5, 1, "lz4")[0])
    for rank, table, offsets in got:
        assert table == want
        assert offsets == [int(x) for x in np.concatenate([[0], np.cumsum(want)[:-1]])]


def _payload_worker(rank, world, port, nchunks, q):
    import ctypes as C
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import DATASETS, orc_compress, orc_decompress
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _multigpu()
    O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    O.orc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int]
    O.orc_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lo, hi = m.chunk_range(nchunks, world, rank)
    rows, cb, sizes = [], [], []
    for c in range(lo, hi):
        n = 20000 + 1000 * c
        data = DATASETS["bench19" if c % 2 == 0 else "randwalk"](n)
        r, ch = orc_compress(O, data, 8, 5, 1, "lz4")
        row = torch.zeros(n + 64, dtype=torch.uint8); row[:r] = torch.from_numpy(ch[:r].copy())
        rows.append(row); cb.append(r); sizes.append(n)
    table, offsets = m.gather_cbytes(cb, nchunks)
    packed = m.pack_local(rows, cb)
    everyone, off_all = m.gather_payload(packed, table, nchunks)                 # all-gather-v
    at_owner, off_own = m.gather_payload(packed, table, nchunks, dst=world - 1)  # gather-v to one owner
    assert off_all == offsets and off_own == offsets
    assert (at_owner is not None) == (rank == world - 1)
    if at_owner is not None:
        assert torch.equal(at_owner, everyone)
    # the inverse: the owner hands every rank its range back; each chunk decodes from where the table says it lies
    mine, local_off = m.scatter_payload(at_owner, table, nchunks, src=world - 1)
    assert torch.equal(mine, packed)
    ok = True
    for k, c in enumerate(range(lo, hi)):
        ch = mine[local_off[k]:local_off[k] + cb[k]].numpy().copy()
        n = sizes[k]
        r, out = orc_decompress(O, ch, n)
        ok = ok and r == n and np.array_equal(out, DATASETS["bench19" if c % 2 == 0 else "randwalk"](n))
    dist.barrier()
    q.put((rank, everyone.numpy().tobytes(), ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("nchunks", [7, 2, 1])
def test_payload_consolidation_world2(oracle, nchunks):
    """all-gather-v of the compressed chunks (counts from the cbytes table) to everybody and to one owner, and the
    scatter back - ragged ranges, a rank without chunks (nchunks = 1), every byte in chunk order."""
    import torch.multiprocessing as mp
    from helpers import DATASETS, orc_compress
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 17 * nchunks) % 1000
    procs = [ctx.Process(target=_payload_worker, args=(r, world, port, nchunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = b""
    for c in range(nchunks):
        data = DATASETS["bench19" if c % 2 == 0 else "randwalk"](20000 + 1000 * c)
        r, ch = orc_compress(oracle, data, 8, 5, 1, "lz4")
        want += ch[:r].tobytes()
    for rank, blob, ok in got:
        assert ok and blob == want, rank
