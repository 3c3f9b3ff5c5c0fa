"""GPU compress path: chunks are valid c-blosc chunks — the oracle (pinned to the reference) and,
when present, the real reference decode them bit-exactly; headers equal the reference's for the
same parameters; return codes follow tests/test_maxout.c and tests/test_compressor.c."""
import ctypes as C

import numpy as np
import pytest

from helpers import DATASETS, orc_compress, orc_decompress, ref_decompress, header, ptr

pytestmark = pytest.mark.gpu


def _check_roundtrip(pkg, oracle, ref, data, T, clevel, shuffle, cname, blocksize=0):
    r, chunk = pkg.compress(data, T, clevel, shuffle, cname, blocksize)
    assert r > 0, (r, T, data.size, clevel, shuffle, cname)
    assert r <= data.size + 16
    h = header(chunk)
    assert h["cbytes"] == r and h["nbytes"] == data.size
    r2, out = orc_decompress(oracle, chunk, data.size)
    assert r2 == data.size and np.array_equal(out, data), ("oracle cannot read GPU chunk", T, data.size, clevel, shuffle, cname, r2)
    if ref is not None:
        r3, out3 = ref_decompress(ref, chunk, data.size)
        assert r3 == data.size and np.array_equal(out3, data), ("stock c-blosc cannot read GPU chunk", T, data.size, cname, r3)
    r4, out4 = pkg.decompress(chunk, data.size)
    assert r4 == data.size and np.array_equal(out4, data)
    return r, chunk


@pytest.mark.parametrize("cname", [b"lz4", b"blosclz", b"lz4hc", b"zstd", b"zlib"])
@pytest.mark.parametrize("shuffle", [0, 1, 2])
def test_roundtrip_grid(pkg, oracle, ref, cname, shuffle):
    """T x N grid of tests/test_compress_roundtrip.csv (+ leftovers), clevels {1,5,9}."""
    for T in [1, 2, 3, 4, 7, 8, 16, 17, 32]:
        for n in [128, 129, 1000, 4096, 32768, 65536 + 17, 300001, 1 << 20, (1 << 21) + 5, 641091]:
            for dname in ["bench19", "randwalk", "zeros", "smallints"]:
                if n > 400000 and T not in (4, 8) and dname != "bench19":
                    continue
                data = DATASETS[dname](n)
                for clevel in ([1, 5, 9] if n <= 300001 else [5]):
                    _check_roundtrip(pkg, oracle, ref, data, T, clevel, shuffle, cname)


def test_headers_equal_reference_policy(pkg, oracle):
    """Same (clevel, typesize, nbytes, codec) -> same version/flags/typesize/nbytes/blocksize bytes as
    the reference writes (blosc.c:1148-1247, compute_blocksize :962-1060)."""
    for cname, codec in [(b"lz4", "lz4"), (b"blosclz", "blosclz")]:
        for T in [1, 4, 8, 16, 32]:
            for n in [1000, 40000, 300001, 1 << 20, (1 << 22) + 12]:
                for clevel in range(1, 10):
                    data = DATASETS["bench19"](n)
                    r, chunk = pkg.compress(data, T, clevel, 1, cname)
                    ro, ochunk = orc_compress(oracle, data, T, clevel, 1, codec)
                    assert r > 0 and ro > 0
                    a, b = chunk[:12].copy(), ochunk[:12].copy()
                    a[2] &= 0xFD; b[2] &= 0xFD      # MEMCPYED depends on how well each ENCODER did, not on policy
                    assert np.array_equal(a, b), (cname, T, n, clevel, header(chunk), header(ochunk))


def test_ratio_close_to_reference(pkg, oracle):
    """Encoders differ, ratios should not collapse: LZ4 within 25% of the reference algorithm's size on
    the bench19 / float64 inputs, BloscLZ within 50% (its far-distance matches cost 4 bytes
    and the reference's encoder is tuned for that; ours shares the LZ4 match finder).  Printed for the record."""
    for cname, codec in [(b"lz4", "lz4"), (b"blosclz", "blosclz")]:
        for dname, T, shuffle in [("bench19", 8, 1), ("bench19", 4, 2), ("linspace", 8, 1), ("randwalk", 8, 1), ("arange", 4, 1)]:
            data = DATASETS[dname](1 << 22)
            r, _ = pkg.compress(data, T, 5, shuffle, cname)
            ro, _ = orc_compress(oracle, data, T, 5, shuffle, codec)
            print(f"ratio {cname.decode():8s} {dname:9s} T={T} shuffle={shuffle}: gpu {data.size / r:8.2f}  reference {data.size / ro:8.2f}")
            assert r <= ro * (1.25 if cname == b"lz4" else 1.5) + 64, (cname, dname, r, ro)


def test_return_codes_maxout(pkg, lib):
    """tests/test_maxout.c:26-144 and tests/test_compressor.c:232-289."""
    data = DATASETS["random"](1000)
    out = np.zeros(2000, np.uint8)
    # destsize = n + 15 on incompressible input -> 0 ; n + 16 -> n + 16 ; dest < 16 -> 0
    assert lib.blosc_compress_ctx(5, 1, 1, 1000, ptr(data), ptr(out), 1000 + 15, b"lz4", 0, 1) == 0
    assert lib.blosc_compress_ctx(5, 1, 1, 1000, ptr(data), ptr(out), 1000 + 16, b"lz4", 0, 1) == 1016
    assert lib.blosc_compress_ctx(5, 1, 1, 1000, ptr(data), ptr(out), 15, b"lz4", 0, 1) == 0
    assert lib.blosc_compress_ctx(5, 1, 1, 1000, ptr(data), ptr(out), 2000, b"lz4", 0, 1) == 1016
    # invalid parameters -> -10 ; unknown / not-built codec -> -5
    assert lib.blosc_compress_ctx(10, 1, 4, 1000, ptr(data), ptr(out), 2000, b"lz4", 0, 1) == -10
    assert lib.blosc_compress_ctx(5, 3, 4, 1000, ptr(data), ptr(out), 2000, b"lz4", 0, 1) == -10
    assert lib.blosc_compress_ctx(5, 1, 0, 1000, ptr(data), ptr(out), 2000, b"lz4", 0, 1) == -10
    assert lib.blosc_compress_ctx(5, 1, 4, 1000, ptr(data), ptr(out), 2000, b"snappy", 0, 1) == -5
    # empty buffer -> 16 ; 1..15 byte buffers -> n + 16
    assert lib.blosc_compress_ctx(5, 1, 4, 0, ptr(data), ptr(out), 2000, b"blosclz", 0, 1) == 16
    for n in range(1, 16):
        assert lib.blosc_compress_ctx(5, 1, 4, n, ptr(data), ptr(out), 2000, b"blosclz", 0, 1) == n + 16
    # typesize > 255 is treated as 1 (blosc.c:1117-1120)
    d2 = DATASETS["bench19"](100000); o2 = np.zeros(d2.size + 16, np.uint8)
    r = lib.blosc_compress_ctx(5, 1, 300, d2.size, ptr(d2), ptr(o2), o2.size, b"lz4", 0, 1)
    assert r > 0 and o2[3] == 1


def test_global_api_and_env(pkg, lib, oracle, monkeypatch):
    """blosc_compress + globals + environment overrides (tests/test_compressor.c:26-229)."""
    data = DATASETS["arange"](8 * 100000)
    out = np.zeros(data.size + 16, np.uint8)
    lib.blosc_init()
    assert lib.blosc_set_compressor(b"lz4") == 1
    r1 = lib.blosc_compress(5, 1, 8, data.size, ptr(data), ptr(out), out.size)
    assert r1 > 0 and (out[2] >> 5) == 1
    monkeypatch.setenv("BLOSC_COMPRESSOR", "blosclz")
    r2 = lib.blosc_compress(5, 1, 8, data.size, ptr(data), ptr(out), out.size)
    assert r2 > 0 and (out[2] >> 5) == 0
    monkeypatch.delenv("BLOSC_COMPRESSOR")
    monkeypatch.setenv("BLOSC_SHUFFLE", "NOSHUFFLE")
    r3 = lib.blosc_compress(5, 1, 8, data.size, ptr(data), ptr(out), out.size)
    assert r3 > r2 and (out[2] & 1) == 0          # shuffle helps on this ramp
    monkeypatch.delenv("BLOSC_SHUFFLE")
    monkeypatch.setenv("BLOSC_CLEVEL", "0")
    assert lib.blosc_compress(5, 1, 8, data.size, ptr(data), ptr(out), out.size) == data.size + 16
    monkeypatch.delenv("BLOSC_CLEVEL")
    monkeypatch.setenv("BLOSC_TYPESIZE", "4")
    assert lib.blosc_compress(5, 1, 8, data.size, ptr(data), ptr(out), out.size) > 0 and out[3] == 4
    monkeypatch.delenv("BLOSC_TYPESIZE")
    monkeypatch.setenv("BLOSC_SPLITMODE", "NEVER")
    assert lib.blosc_compress(5, 1, 8, data.size, ptr(data), ptr(out), out.size) > 0 and (out[2] & 0x10)
    monkeypatch.delenv("BLOSC_SPLITMODE")
    lib.blosc_set_splitmode(4)
    lib.blosc_set_blocksize(65536)
    assert lib.blosc_compress(5, 1, 8, data.size, ptr(data), ptr(out), out.size) > 0
    # a forced blocksize still goes through the split enlargement (blosc.c:1031-1047): ask the oracle
    want = oracle.orc_compute_blocksize(5, 8, data.size, 65536, 0, 4)
    assert int(out[8:12].view("<i4")[0]) == want
    lib.blosc_set_blocksize(0)
    back = np.zeros(data.size, np.uint8)
    assert lib.blosc_decompress(ptr(out), ptr(back), back.size) == data.size and np.array_equal(back, data)
    lib.blosc_set_compressor(b"blosclz")
    lib.blosc_destroy()


def test_one_lz4_stream_beyond_256_mib(pkg, lib, ref, monkeypatch):
    """ONE stream of 266 MiB: a forced blocksize as large as the chunk and BLOSC_SPLITMODE=NEVER make the whole chunk a single LZ4 stream (with the
    default split the reference caps a block at 1 MiB, blosc/blosc.c:1031-1047).  Round 6's parallel LZ4 writer carried a candidate's POSITION
    next to a 4-bit count in one word - fine below 2^28, silently wrong above: the chunk came out with the right size and return value and could
    not be read back (scripts/dbg_big_stream.py).  Matches of the stream's last megabytes must decode, here and in the reference."""
    if ref is None:
        pytest.skip("needs the reference (oracle/_ref) as reader")
    rng = np.random.default_rng(1)
    head = rng.integers(0, 256, 258 << 20, dtype=np.uint8)                           # match-less: crossed in long strides
    tail = np.ascontiguousarray(DATASETS["bench19"](8 << 20).reshape(-1, 8).T).reshape(-1)      # planes of the bench data as one byte stream: matches a few KiB back
    data = np.concatenate([head, tail])
    n = data.size
    out = np.zeros(n + 16, np.uint8)
    monkeypatch.setenv("BLOSC_SPLITMODE", "NEVER")
    lib.blosc_init()
    try:
        assert lib.blosc_set_compressor(b"lz4") == 1
        lib.blosc_set_blocksize(n)
        for clevel in (5, 9):                                                        # (both strides of the parallel parse)
            cb = lib.blosc_compress(clevel, 0, 8, n, ptr(data), ptr(out), n + 16)
            assert 0 < cb < n - (4 << 20), cb                                        # the tail did compress
            assert int(out[8:12].view("<i4")[0]) == n and (out[2] & 0x10)            # one block, not split
            r, back = ref_decompress(ref, out[:cb], n)
            assert r == n and np.array_equal(back, data), ("the reference cannot read the chunk", clevel)
            mine = np.zeros(n, np.uint8)
            assert lib.blosc_decompress(ptr(out), ptr(mine), n) == n and np.array_equal(mine, data)
        # ... and the other direction: the reference's own single stream of that size, decoded here
        ref.blosc_init()
        ref.blosc_set_compressor(b"lz4"); ref.blosc_set_blocksize(n)
        ref.blosc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        cb = ref.blosc_compress(5, 0, 8, n, ptr(data), ptr(out), n + 16)
        ref.blosc_set_blocksize(0); ref.blosc_destroy()
        assert 0 < cb < n and int(out[8:12].view("<i4")[0]) == n and (out[2] & 0x10)
        mine = np.zeros(n, np.uint8)
        assert lib.blosc_decompress(ptr(out), ptr(mine), n) == n and np.array_equal(mine, data), "a reference-written stream of 266 MiB"
    finally:
        lib.blosc_set_blocksize(0)
        lib.blosc_set_compressor(b"blosclz")
        lib.blosc_destroy()


@pytest.mark.parametrize("cname,shuffle", [(b"blosclz", 1), (b"lz4", 2), (b"lz4hc", 1), (b"zstd", 1), (b"zlib", 2)])
def test_one_stream_beyond_256_mib_of_the_other_writers(pkg, lib, ref, monkeypatch, cname, shuffle):
    """The same question as above put to the other writers, the decoders and the fused filters: a block of 266 MiB, not split, under a byte or bit
    shuffle - eight planes of 33 MiB as ONE stream whose last plane lies beyond 2^27 * 1.75, every plane with its matches.  Written here and read
    by the reference, written by the reference and read here (scripts/dbg_big_stream_all.py is the full grid: 5 codecs x 3 filters x 3 pairings,
    profiles/r06zl_*)."""
    if ref is None:
        pytest.skip("needs the reference (oracle/_ref) as writer and reader")
    n = 266 << 20
    data = DATASETS["bench19"](n)
    out = np.zeros(n + 16, np.uint8)
    clevel = 5 if cname in (b"blosclz", b"lz4") else 1
    monkeypatch.setenv("BLOSC_SPLITMODE", "NEVER")
    for L in (lib, ref):
        L.blosc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        L.blosc_set_blocksize.argtypes = [C.c_size_t]
    lib.blosc_init()
    try:
        assert lib.blosc_set_compressor(cname) >= 0
        lib.blosc_set_blocksize(n)
        cb = lib.blosc_compress(clevel, shuffle, 8, n, ptr(data), ptr(out), n + 16)
        lib.blosc_set_blocksize(0)
        assert 0 < cb < n // 8, cb
        assert int(out[8:12].view("<i4")[0]) == n and (out[2] & 0x10)                # one block, not split
        r, back = ref_decompress(ref, out[:cb], n)
        assert r == n and np.array_equal(back, data), ("the reference cannot read the chunk", cname)
        ref.blosc_init()
        ref.blosc_set_compressor(cname); ref.blosc_set_blocksize(n)
        cb = ref.blosc_compress(clevel, shuffle, 8, n, ptr(data), ptr(out), n + 16)
        ref.blosc_set_blocksize(0); ref.blosc_destroy()
        assert 0 < cb < n // 8 and int(out[8:12].view("<i4")[0]) == n and (out[2] & 0x10)
        mine = np.zeros(n, np.uint8)
        assert lib.blosc_decompress(ptr(out), ptr(mine), n) == n and np.array_equal(mine, data), ("a reference-written stream of 266 MiB", cname)
    finally:
        lib.blosc_set_blocksize(0)
        lib.blosc_set_compressor(b"blosclz")
        lib.blosc_destroy()
