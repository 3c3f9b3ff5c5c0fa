"""GPU: bitshuffle chunks of typesize 1 / 2 / 4 / 8 (8: round 5, two bytes of every bit row per lane) are bit-unshuffled INSIDE the decode kernel (round 4: the wave that completes a block's last
stream transposes the block, k_decode.hip: bitunshuffle_block_wave) - no k_bitunshuffle pass.  Reference-written chunks (oracle = pinned to
the reference), every corner rule of blosc/shuffle.c:393-443: element counts that are not multiples of 8 (whole block copied), of 32 and of
2048 / 1024 (the wave's partial passes), trailing bytes, leftover blocks, blocks smaller than the type size; typesize 3 (and every other
odd size) keeps the stand-alone kernels (and must still be right).  The kernel profile says which path ran."""
import numpy as np
import pytest

from helpers import DATASETS, orc_compress

pytestmark = pytest.mark.gpu


def _launches(pkg, name):
    return pkg.profile_get(name)[1]


@pytest.mark.parametrize("codec", ["lz4", "blosclz"])
@pytest.mark.parametrize("T", [1, 2, 4, 8, 3])
def test_bitshuffle_chunks_decode_like_the_reference(pkg, lib, oracle, codec, T):
    rng = np.random.default_rng(100 + T)
    sizes = [8 << 20, (4 << 20) + 40, 641091, 2048 * T * 5 + 32 * T * 3 + 8 * T + (T - 1), 1024 * T * 3 + 16 * T * 5 + 8 * T, 1000 * T, 31 * T, 7, T - 1 if T > 1 else 1, 65536 * T + 8 * T]
    for n in sizes:
        if n <= 0:
            continue
        for dname in ("bench19", "smallints"):
            data = DATASETS[dname](n)
            for blocksize in (0, 16384 if n > 40000 else 0):
                r, chunk = orc_compress(oracle, data, T, 5, 2, codec, blocksize=blocksize)
                assert r > 0
                lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
                got_r, got = pkg.decompress(chunk, n)
                lib.blosc_gpu_profile(0)
                assert got_r == n and np.array_equal(got, data), (codec, T, n, dname, blocksize)
                fused = T in (1, 2, 4, 8)
                memcpyed = bool(chunk[2] & 2)
                if not memcpyed and n >= T:
                    assert (_launches(pkg, "k_bitunshuffle") == 0) == fused, (codec, T, n, _launches(pkg, "k_bitunshuffle"))


def test_mixed_batch_of_fused_and_stand_alone_bit_chunks(pkg, oracle):
    """typesize 4 / 8 / 2 (fused) and typesize 3 / 6 (stand-alone kernels) bitshuffle chunks in ONE batch: the stand-alone pass must leave the fused chunks alone."""
    import ctypes as C
    datas, chunks = [], []
    for k, T in enumerate([4, 3, 8, 6, 2, 8, 3]):
        n = (1 << 20) + 48 * k
        d = DATASETS["bench19" if k % 2 == 0 else "smallints"](n)
        r, ch = orc_compress(oracle, d, T, 5, 2, "lz4")
        datas.append(d); chunks.append(ch)
    outs = [np.zeros(d.size, np.uint8) for d in datas]
    n = len(chunks)
    L = pkg.load()
    src = (C.c_void_p * n)(*[c.ctypes.data for c in chunks]); dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    ssz = (C.c_size_t * n)(*[c.size for c in chunks]); dsz = (C.c_size_t * n)(*[o.size for o in outs]); res = (C.c_int * n)()
    assert L.blosc_gpu_decompress_batch_host(n, src, ssz, dst, dsz, res) == 0
    for k in range(n):
        assert res[k] == datas[k].size and np.array_equal(outs[k], datas[k]), k


@pytest.mark.parametrize("codec", ["lz4", "blosclz", "zstd"])
@pytest.mark.parametrize("T", [1, 2, 4, 8, 3])
def test_bitshuffle_chunks_written_here_are_read_by_the_reference(pkg, lib, oracle, codec, T):
    """The compress side: bitshuffle of typesize 1 / 2 / 4 / 8 is a task of the encode kernel (enc_shuffle.h: bitshuffle_block_wave_T), no
    k_bitshuffle pass; the chunks must decode bit-exactly with the oracle (= the reference's reader) and with our own decoder."""
    from helpers import orc_decompress
    sizes = [8 << 20, (4 << 20) + 40, 641091, 2048 * T * 5 + 32 * T * 3 + 8 * T + (T - 1), 1024 * T * 3 + 16 * T * 5 + 8 * T, 1000 * T, 31 * T, 7, 65536 * T + 8 * T]
    for n in sizes:
        for dname in ("bench19", "smallints"):
            data = DATASETS[dname](n)
            lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
            r, chunk = pkg.compress(data, T, 5 if codec != "zstd" else 3, 2, codec.encode())
            lib.blosc_gpu_profile(0)
            assert r > 0, (codec, T, n, r)
            memcpyed = bool(chunk[2] & 2)
            if not memcpyed:
                assert (_launches(pkg, "k_bitshuffle") == 0) == (T in (1, 2, 4, 8)), (codec, T, n)
            if codec != "zstd":                                    # (the plain-C oracle reads LZ4 / BloscLZ; Zstd chunks go through our decoder below and the reference in test_gpu_zstd.py)
                ro, back = orc_decompress(oracle, chunk, n)
                assert ro == n and np.array_equal(back, data), (codec, T, n, dname)
            rg, got = pkg.decompress(chunk, n)
            assert rg == n and np.array_equal(got, data), (codec, T, n, dname)
