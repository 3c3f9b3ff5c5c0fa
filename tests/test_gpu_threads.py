"""GPU: re-entrancy of the `_ctx` calls.  blosc_compress_ctx / blosc_decompress_ctx / blosc_getitem are what multi-threaded callers
of the reference use - every call builds a context of its own and callers run side by side (blosc/blosc.c:1288-1305, :1560-1572,
:1618-1690).  The engine keeps a small pool of contexts (engine.hip: CtxGuard; BLOSC_AMD_CONTEXTS) with their own workspaces and,
for host buffers, their own streams: threads calling at the same time must get exactly what they get one after the other."""
import ctypes as C
import threading

import numpy as np
import pytest

from helpers import DATASETS, orc_decompress, ptr, ref_compress

pytestmark = pytest.mark.gpu

JOBS = [  # (data set, bytes, typesize, shuffle, codec, clevel)
    ("bench19", 3 << 20, 8, 1, b"lz4", 5),
    ("linspace", (2 << 20) + 40, 8, 1, b"blosclz", 5),
    ("smallints", 1 << 20, 4, 2, b"lz4", 9),
    ("randwalk", 1500000, 8, 1, b"zstd", 3),
    ("bench19", 1 << 20, 16, 1, b"lz4", 5),
    ("smallints", 700001, 2, 1, b"zlib", 5),
    ("zeros", 1 << 20, 4, 1, b"lz4hc", 9),
    ("random", 300000, 1, 0, b"lz4", 5),
]


def _worker(lib, oracle, ref, job, rounds, errors, tid):
    try:
        dname, n, T, shuffle, cname, clevel = job
        data = DATASETS[dname](n)
        stock = None
        if ref is not None:
            r, stock = ref_compress(ref, data, T, clevel, shuffle, cname)
            assert r > 0
        first = None
        for it in range(rounds):
            dst = np.full(n + 16 + 64, 0xEE, np.uint8)
            r = lib.blosc_compress_ctx(clevel, shuffle, T, n, ptr(data), ptr(dst), n + 16, cname, 0, 1)
            assert 0 < r <= n + 16 and np.all(dst[n + 16:] == 0xEE), (tid, it, r)
            chunk = dst[:r].copy()
            if first is None:
                first = chunk
                r2, out = orc_decompress(oracle, chunk, n)
                assert r2 == n and np.array_equal(out, data), (tid, "the oracle cannot read it")
            elif cname in (b"lz4", b"blosclz") or (cname == b"zstd" and clevel <= 5):      # (the search modes - lz4hc, zlib, zstd from clevel 6 - let lanes race for a bucket slot: valid either way; include/blosc.h says so)
                assert np.array_equal(chunk, first), (tid, it, "the same call gave different bytes")      # queue order never changes the bytes of a stream
            back = np.full(n + 64, 0xEE, np.uint8)
            src = stock if (stock is not None and it % 2) else chunk
            assert lib.blosc_decompress_ctx(ptr(src), ptr(back), n, 1) == n, (tid, it)
            assert np.array_equal(back[:n], data) and np.all(back[n:] == 0xEE), (tid, it)
            nitems = min(1000, n // T)
            start = (n // T - nitems) // 2
            item = np.full(nitems * T + 16, 0xEE, np.uint8)
            assert lib.blosc_getitem(ptr(src), start, nitems, ptr(item)) == nitems * T, (tid, it)
            assert np.array_equal(item[:nitems * T], data[start * T:(start + nitems) * T]) and np.all(item[nitems * T:] == 0xEE), (tid, it)
    except BaseException as e:      # noqa: BLE001 - reported by the main thread
        errors.append((tid, repr(e)))


@pytest.mark.parametrize("nthreads", [2, 8, 12])
def test_ctx_calls_from_many_threads(lib, oracle, ref, nthreads):
    """2 threads (fewer than contexts), 8 (more than the default 4 contexts) and 12 (more than the pool can ever hold): host buffers,
    every codec, typesizes 1 .. 16; each thread checks its own results against the oracle, the reference's chunks and its first call."""
    errors = []
    threads = [threading.Thread(target=_worker, args=(lib, oracle, ref, JOBS[t % len(JOBS)], 4, errors, t)) for t in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_device_batches_from_two_threads(pkg, lib, oracle):
    """The batched, device-resident extension from two threads at once (each batch on the null stream, as the callers name none):
    two workspaces, results identical to the serial ones."""
    import torch
    dev = torch.device("cuda:0")
    n, nch = 2 << 20, 6
    outs = {}
    errors = []

    def run(tid, dname, T):
        try:
            data = DATASETS[dname](n)
            src = torch.from_numpy(data).to(dev).unsqueeze(0).repeat(nch, 1).contiguous()
            comp = torch.zeros((nch, n + 16), dtype=torch.uint8, device=dev)
            back = torch.zeros((nch, n), dtype=torch.uint8, device=dev)
            for it in range(3):
                b = pkg.DeviceBatch([src[i].data_ptr() for i in range(nch)], [n] * nch, [comp[i].data_ptr() for i in range(nch)], [n + 16] * nch)
                assert b.compress(T, 5, 1, b"lz4") == 0
                cb = b.results()
                assert all(0 < c < n for c in cb) and len(set(cb)) == 1, cb
                d = pkg.DeviceBatch([comp[i].data_ptr() for i in range(nch)], cb, [back[i].data_ptr() for i in range(nch)], [n] * nch)
                assert d.decompress() == 0 and d.results() == [n] * nch
                assert bool((back == src).all()), (tid, it)
            chunk = comp[0, :cb[0]].cpu().numpy()
            r, out = orc_decompress(oracle, chunk, n)
            assert r == n and np.array_equal(out, data)
            outs[tid] = cb[0]
        except BaseException as e:      # noqa: BLE001
            errors.append((tid, repr(e)))

    th = [threading.Thread(target=run, args=(0, "bench19", 8)), threading.Thread(target=run, args=(1, "smallints", 4))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(outs) == 2
