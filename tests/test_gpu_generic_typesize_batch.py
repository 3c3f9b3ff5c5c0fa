"""GPU: byte (un)shuffle of the "other" typesizes (3 ... 32 except 4 / 8 / 16) inside the codec kernels, at batch scale.  The shuffle tasks of these
typesizes are the slow LDS-tile form, so on incompressible data the waves whose stream's block is not shuffled yet claim shuffle tasks themselves
(k_encode.hip: the per-XCD shuffle list) - a path that only a batch with thousands of blocks in flight exercises.  Every chunk written here must
be read back bit-exactly by the oracle (= the reference's reader) and by our decoder; reference-written chunks of the same data must decode too;
the kernel profile says that no stand-alone filter pass ran."""
import ctypes as C

import numpy as np
import pytest

from helpers import DATASETS, orc_compress, orc_decompress, ref_compress, ref_decompress

pytestmark = pytest.mark.gpu


def _batch(L, fn, n, src, ssz, dst, dsz, *front):
    s = (C.c_void_p * n)(*[a.ctypes.data for a in src]); d = (C.c_void_p * n)(*[a.ctypes.data for a in dst])
    ss = (C.c_size_t * n)(*ssz); ds = (C.c_size_t * n)(*dsz); res = (C.c_int * n)()
    assert fn(*front, n, s, ss, d, ds, res) == 0
    return list(res)


@pytest.mark.parametrize("T", [3, 6, 12, 24, 31])
def test_many_chunk_batches_of_other_typesizes(pkg, lib, oracle, T):
    L = pkg.load()
    nchunks, n = 48, (4 << 20) + 8 * T + 5
    rng = np.random.default_rng(T)
    kinds = ["random", "bench19", "linspace", "random"]            # incompressible chunks next to compressible ones
    datas = [DATASETS[kinds[k % 4]](n) if kinds[k % 4] != "random" else rng.integers(0, 256, n, dtype=np.uint8) for k in range(nchunks)]
    comps = [np.zeros(n + 16, np.uint8) for _ in range(nchunks)]
    for cname in (b"lz4", b"blosclz"):
        lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
        cb = _batch(L, L.blosc_gpu_compress_batch_host, nchunks, datas, [n] * nchunks, comps, [n + 16] * nchunks, 5, 1, T, cname, 0)
        assert all(0 < c <= n + 16 for c in cb), cb[:8]
        outs = [np.full(n, 0xEE, np.uint8) for _ in range(nchunks)]
        res = _batch(L, L.blosc_gpu_decompress_batch_host, nchunks, comps, cb, outs, [n] * nchunks)
        lib.blosc_gpu_profile(0)
        assert pkg.profile_get("k_shuffle")[1] == 0 and pkg.profile_get("k_unshuffle")[1] == 0, "a stand-alone filter pass ran"
        for k in range(nchunks):
            assert res[k] == n and np.array_equal(outs[k], datas[k]), (T, cname, k)
        for k in (0, 1, 2, nchunks - 1):                              # the reference's reader on what was written here
            ro, back = orc_decompress(oracle, comps[k][:cb[k]], n)
            assert ro == n and np.array_equal(back, datas[k]), (T, cname, k)
    # reference-written chunks of the same data through the decode kernel's own unshuffle
    refc = []
    for k in range(8):
        r, ch = orc_compress(oracle, datas[k], T, 5, 1, "lz4")
        assert r > 0
        refc.append(ch)
    outs = [np.full(n, 0xEE, np.uint8) for _ in range(8)]
    res = _batch(L, L.blosc_gpu_decompress_batch_host, 8, refc, [c.size for c in refc], outs, [n] * 8)
    for k in range(8):
        assert res[k] == n and np.array_equal(outs[k], datas[k]), (T, "stock", k)


def test_batches_of_one_geometry_after_another(pkg, lib, oracle, ref):
    """Late in round 6 a workspace keeps the block table and the task queues of its last call per direction on the device (engine.hip: TableCache) and a
    call of the same geometry takes them.  Batches back to back: the same shape with OTHER content in OTHER buffers (a hit), a ragged one, another
    typesize, one chunk less, the first shape again - every chunk must read back everywhere, and the reference's chunks of the same shapes must decode
    through the tables a batch of OUR chunks left behind (equal geometry, other compressed sizes)."""
    L = pkg.load()
    rng = np.random.default_rng(11)
    n = (2 << 20) + 4096

    def run(datas, T, shuffle, cname):
        k = len(datas)
        comps = [np.zeros(d.size + 16, np.uint8) for d in datas]
        cb = _batch(L, L.blosc_gpu_compress_batch_host, k, datas, [d.size for d in datas], comps, [d.size + 16 for d in datas], 5, shuffle, T, cname, 0)
        assert all(0 < c <= d.size + 16 for c, d in zip(cb, datas)), cb[:8]
        outs = [np.full(d.size, 0xEE, np.uint8) for d in datas]
        res = _batch(L, L.blosc_gpu_decompress_batch_host, k, comps, cb, outs, [d.size for d in datas])
        for i in range(k):
            assert res[i] == datas[i].size and np.array_equal(outs[i], datas[i]), (T, shuffle, cname, i)
        for i in (0, k // 2, k - 1):
            ro, back = (ref_decompress(ref, comps[i][:cb[i]], datas[i].size) if ref is not None else orc_decompress(oracle, comps[i][:cb[i]], datas[i].size))
            assert ro == datas[i].size and np.array_equal(back, datas[i]), ("the reference's reader", T, shuffle, cname, i)
        return comps, cb

    a = [DATASETS["bench19"](n) for _ in range(24)]
    b = [DATASETS[("linspace", "randwalk", "bench19")[i % 3]](n) if i % 4 else rng.integers(0, 256, n, dtype=np.uint8) for i in range(24)]
    ragged = [DATASETS["bench19"](n - 997 * i) for i in range(24)]
    for cname in (b"lz4", b"blosclz", b"zstd"):
        run(a, 8, 1, cname); run(a, 8, 1, cname); run(b, 8, 1, cname); run(ragged, 8, 1, cname); run(b, 4, 1, cname); run(b, 8, 2, cname)
        run(b[:23], 8, 1, cname); run(a, 8, 1, cname); run(b, 8, 0, cname); run(a, 8, 1, cname)
        # reference-written chunks of the same geometry as the last batch, then of another one
        for datas, T in ((a, 8), (b, 8), (b, 4)):
            refc = []
            for d in datas[:12]:
                r, ch = (ref_compress(ref, d, T, 5, 1, cname) if ref is not None else orc_compress(oracle, d, T, 5, 1, cname.decode()))
                assert r > 0
                refc.append(np.ascontiguousarray(ch[:r]))
            outs = [np.full(n, 0xEE, np.uint8) for _ in range(12)]
            res = _batch(L, L.blosc_gpu_decompress_batch_host, 12, refc, [c.size for c in refc], outs, [n] * 12)
            for i in range(12):
                assert res[i] == n and np.array_equal(outs[i], datas[i]), ("reference-written", cname, T, i)
