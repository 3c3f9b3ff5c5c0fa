"""GPU: byte (un)shuffle of the "other" typesizes (3 ... 32 except 4 / 8 / 16) inside the codec kernels, at batch scale.  The shuffle tasks of these
typesizes are the slow LDS-tile form, so on incompressible data the waves whose stream's block is not shuffled yet claim shuffle tasks themselves
(k_encode.hip: the per-XCD shuffle list) - a path that only a batch with thousands of blocks in flight exercises.  Every chunk written here must
be read back bit-exactly by the oracle (= the reference's reader) and by our decoder; reference-written chunks of the same data must decode too;
the kernel profile says that no stand-alone filter pass ran."""
import ctypes as C

import numpy as np
import pytest

from helpers import DATASETS, orc_compress, orc_decompress

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
