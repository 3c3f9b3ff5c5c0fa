"""CPU: oracle/zstd_oracle.c (Zstd frame decoder restated from the format) pinned against the reference:
the compat/*zstd*.cdata golden vectors, committed chunks written by the real reference
(tests/golden/ref_zstd_chunks.npz, made by make_ref_zstd_chunks.py) and, where oracle/_ref exists, a sweep of
freshly written chunks incl. getitem.  The GPU decoder (k_zstd.hip) is checked against this oracle and the same fixtures in
tests/test_gpu_zstd.py."""
import glob
import os

import numpy as np
import pytest

from helpers import DATASETS, orc_decompress, ptr, ref_compress

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("fname", sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "compat", "*zstd*.cdata"))))
def test_compat_zstd_vectors(oracle, fname):
    chunk = np.fromfile(os.path.join(GOLDEN, "compat", fname), np.uint8)
    r, out = orc_decompress(oracle, chunk, 4000000)
    assert r == 4000000 and np.array_equal(out.view("<i4"), np.arange(10**6, dtype="<i4"))


def test_committed_reference_chunks(oracle):
    z = np.load(os.path.join(GOLDEN, "ref_zstd_chunks.npz"))
    meta = [m.split(",") for m in z["meta"]]
    assert len(meta) >= 24
    for k, (dname, n, T, clevel, shuffle, bs) in enumerate(meta):
        n = int(n)
        r, out = orc_decompress(oracle, z[f"c{k}"], n)
        assert r == n and np.array_equal(out, DATASETS[dname](n)), meta[k]


def test_truncated_and_corrupt_frames_fail_cleanly(oracle):
    z = np.load(os.path.join(GOLDEN, "ref_zstd_chunks.npz"))
    chunk = z["c0"]; n = int(z["meta"][0].split(",")[1])
    rng = np.random.default_rng(3)
    for trial in range(300):
        c = chunk.copy()
        pos = int(rng.integers(16, c.size)); c[pos] ^= 1 << int(rng.integers(0, 8))
        out = np.zeros(n, np.uint8)
        r = oracle.orc_decompress(ptr(c), ptr(out), n)       # must return (any code), never crash or overrun
        assert r == n or r < 0


def test_fresh_reference_chunks(oracle, ref):
    """sweep against the real reference (only where oracle/_ref is built)"""
    for dname in ("bench19", "linspace", "randwalk", "smallints"):
        for clevel in (1, 4, 9):
            for T, shuffle in ((8, 1), (4, 2), (2, 1), (1, 0)):
                n = 1 << 19
                data = DATASETS[dname](n)
                r, chunk = ref_compress(ref, data, T, clevel, shuffle, b"zstd")
                assert r > 0
                d, out = orc_decompress(oracle, chunk, n)
                assert d == n and np.array_equal(out, data), (dname, clevel, T, shuffle)
                # getitem on the same chunk (blosc.c:1574-1703)
                start, nitems = 1000, 5000
                got = np.zeros(nitems * T, np.uint8)
                assert oracle.orc_getitem(ptr(chunk), start, nitems, ptr(got)) == nitems * T
                assert np.array_equal(got, data[start * T:(start + nitems) * T])


def _direct_frames(ref):
    """frames written by the reference's own ZSTD_compress (exported by oracle/_ref): sizes 1 .. 700 000 (several
    blocks per frame), levels -5 .. 22, constant / random / low-entropy / text / random-walk / periodic data"""
    import ctypes as C
    ref.ZSTD_compress.restype = C.c_size_t
    ref.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    ref.ZSTD_compressBound.restype = C.c_size_t
    ref.ZSTD_compressBound.argtypes = [C.c_size_t]
    ref.ZSTD_isError.argtypes = [C.c_size_t]
    rng = np.random.default_rng(0)

    def gens(n):
        yield np.full(n, 7, np.uint8)
        yield rng.integers(0, 256, n, dtype=np.uint8)
        yield rng.integers(0, 4, n, dtype=np.uint8)
        yield np.frombuffer((b"the quick brown fox jumps over the lazy dog. " * (n // 44 + 1))[:n], np.uint8).copy()
        yield np.cumsum(rng.integers(-2, 3, n // 4 + 1)).astype("<i4").view(np.uint8)[:n].copy()
        p = np.tile(rng.integers(0, 256, 97, dtype=np.uint8), n // 97 + 1)[:n].copy(); p[::501] ^= 1
        yield p

    for n in [1, 2, 3, 5, 16, 63, 64, 255, 256, 257, 1000, 4096, 5000, 65536, 131072, 131073, 300000, 700000]:
        for data in gens(n):
            for lvl in (1, 3, 5, 9, 15, 19, 22, -5):
                cap = ref.ZSTD_compressBound(n); buf = np.empty(cap, np.uint8)
                r = ref.ZSTD_compress(ptr(buf), cap, ptr(data), n, lvl)
                assert not ref.ZSTD_isError(r)
                yield buf[:r].copy(), data


def test_direct_reference_frames(oracle, ref):
    import ctypes as C
    oracle.orc_zstd_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    count = 0
    for frame, data in _direct_frames(ref):
        out = np.zeros(data.size, np.uint8)
        assert oracle.orc_zstd_decompress(ptr(frame), frame.size, ptr(out), data.size) == data.size
        assert np.array_equal(out, data)
        count += 1
    assert count == 864


def _checksum_frames(ref):
    """frames with a Content_Checksum, written by the reference's advanced API"""
    import ctypes as C
    ref.ZSTD_createCCtx.restype = C.c_void_p
    ref.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    ref.ZSTD_CCtx_setParameter.restype = C.c_size_t
    ref.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    ref.ZSTD_compress2.restype = C.c_size_t
    ref.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    ref.ZSTD_isError.argtypes = [C.c_size_t]
    out = []
    for dname, n in [("bench19", 100000), ("linspace", 131072), ("random", 5000), ("zeros", 31), ("smallints", 300000)]:
        data = DATASETS[dname](n)
        cctx = ref.ZSTD_createCCtx()
        assert not ref.ZSTD_isError(ref.ZSTD_CCtx_setParameter(cctx, 201, 1))       # ZSTD_c_checksumFlag
        assert not ref.ZSTD_isError(ref.ZSTD_CCtx_setParameter(cctx, 100, 3))       # ZSTD_c_compressionLevel
        buf = np.zeros(n + n // 4 + 128, np.uint8)
        r = ref.ZSTD_compress2(cctx, ptr(buf), buf.size, ptr(data), n)
        ref.ZSTD_freeCCtx(cctx)
        assert not ref.ZSTD_isError(r)
        out.append((data, buf[:r].copy()))
    return out


def test_content_checksum_is_verified(oracle, ref):
    """RFC 8878 3.1.1 Content_Checksum (low 32 bits of XXH64): the reference verifies it, so does the oracle."""
    import ctypes as C
    if ref is None:
        pytest.skip("needs oracle/_ref to write frames with a checksum")
    oracle.orc_zstd_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    ref.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    ref.ZSTD_decompress.restype = C.c_size_t
    for data, frame in _checksum_frames(ref):
        assert frame[4] & 4                                                           # checksum flag of the frame header descriptor
        back = np.zeros(data.size + 8, np.uint8)
        assert oracle.orc_zstd_decompress(ptr(frame), frame.size, ptr(back), data.size) == data.size and np.array_equal(back[:data.size], data)
        bad = frame.copy(); bad[-2] ^= 0x10                                           # damage the checksum itself
        d = ref.ZSTD_decompress(ptr(back), data.size, ptr(bad), bad.size)
        assert ref.ZSTD_isError(d)
        assert oracle.orc_zstd_decompress(ptr(bad), bad.size, ptr(back), data.size) <= 0          # 0 = error (zstd_wrap_decompress, blosc.c:515-522)


def _damage(rng, frame):
    """one damaged copy of a frame: bit flips, byte overwrites, truncation, or a cut-and-splice of its own bytes"""
    c = frame.copy()
    kind = int(rng.integers(0, 5))
    if kind == 0:
        pos = int(rng.integers(0, c.size)); c[pos] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        for pos in rng.integers(0, c.size, int(rng.integers(2, 6))):
            c[pos] = int(rng.integers(0, 256))
    elif kind == 2:
        c = c[: int(rng.integers(1, c.size))].copy()
    elif kind == 3 and c.size > 24:
        a = int(rng.integers(4, c.size - 8)); b = int(rng.integers(4, c.size - 8)); k = int(rng.integers(1, 8))
        c[a:a + k] = frame[b:b + k]
    else:                                       # the header / first block header / section headers: where most decisions sit
        pos = int(rng.integers(4, min(c.size, 24))); c[pos] = int(rng.integers(0, 256))
    return c


def _verdicts(oracle, ref, c, n):
    """(reference accepts, oracle accepts, oracle accepts with the Huffman end-of-stream check off, reference bytes, oracle bytes)"""
    want = np.full(n + 64, 0xA5, np.uint8); got = np.full(n + 64, 0xA5, np.uint8); tmp = np.full(n + 64, 0xA5, np.uint8)
    rr = ref.ZSTD_decompress(ptr(want), n, ptr(c), c.size)
    ro = oracle.orc_zstd_decompress(ptr(c), c.size, ptr(got), n)
    oracle.orc_zstd_set_huf_lenient(1)
    try:
        rl = oracle.orc_zstd_decompress(ptr(c), c.size, ptr(tmp), n)
    finally:
        oracle.orc_zstd_set_huf_lenient(0)
    assert np.all(got[n:] == 0xA5) and np.all(tmp[n:] == 0xA5)           # never a byte behind the output
    ref_ok = (not ref.ZSTD_isError(rr)) and rr == n                      # blosc requires exactly the block size (blosc.c:778-782)
    return ref_ok, ro == n, rl == n, want[:n], got[:n]


def _bind_zstd(oracle, ref):
    import ctypes as C
    oracle.orc_zstd_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    oracle.orc_zstd_set_huf_lenient.argtypes = [C.c_int]
    ref.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    ref.ZSTD_decompress.restype = C.c_size_t
    ref.ZSTD_isError.argtypes = [C.c_size_t]


def test_damaged_frames_verdict_and_bytes_equal_reference(oracle, ref):
    """The hole VERDICT r02 names: the oracle's verdict on DAMAGED frames was only ever compared with the GPU decoder, which
    shares its author.  Here >= 2000 damaged frames (all literal / sequence section modes, one and several blocks) go
    through the reference's own ZSTD_decompress (zstd_decompress.c:1201) and through oracle/zstd_oracle.c.  blosc calls
    ZSTD_decompress with dstCapacity = the block's size and takes any error as failure (zstd_wrap_decompress, blosc.c:515-522).
    Contract checked:
      * whatever the reference rejects, the oracle rejects;
      * whatever both accept has the same bytes;
      * the oracle rejects a frame the reference accepts ONLY for a Huffman literal stream that does not end exactly at its
        first bit: RFC 8878 4.2.2 calls that corruption and the reference's portable loop enforces it (huf_decompress.c:692-693),
        its fast 4-stream loop does not (:873-887) and returns bytes decoded from whatever lies in front of the stream.  Shown by
        switching that one check off in the oracle (orc_zstd_set_huf_lenient): every such frame is then accepted.  This is
        the documented deviation of INTEGRATION.md (stricter on corrupt input, never laxer)."""
    if ref is None:
        pytest.skip("needs oracle/_ref")
    _bind_zstd(oracle, ref)
    rng = np.random.default_rng(20260924)
    frames = [(f, d) for f, d in _direct_frames(ref) if 40 <= f.size <= 40000]
    pick = rng.permutation(len(frames))[:130]
    checked = accepted = stricter = 0
    for k in pick:
        frame, data = frames[int(k)]
        n = data.size
        for trial in range(18):
            c = _damage(rng, frame)
            ref_ok, orc_ok, len_ok, want, got = _verdicts(oracle, ref, c, n)
            if not ref_ok:
                assert not orc_ok, (int(k), trial, frame.size, n)
            elif orc_ok:
                assert np.array_equal(got, want), (int(k), trial)
                accepted += 1
            else:
                assert len_ok, (int(k), trial, frame.size, n)      # the Huffman end-of-stream rule is the only difference
                stricter += 1
            checked += 1
    assert checked >= 2000 and accepted >= 20, (checked, accepted)
    assert stricter * 50 <= checked, (stricter, checked)         # a rarity (12 of 2340 with this seed), not a second decoder


def test_damaged_frames_equal_reference_portable_build(oracle, ref):
    """...and with NO exception against the reference built with its own switch HUF_DISABLE_FAST_DECODE (huf_decompress.c:37-41;
    oracle/Makefile: _ref/libzstd_ref_portable.so - the Huffman loops every non-64-bit-little-endian build of the reference runs):
    the same verdict on every damaged frame, the same bytes on every accepted one."""
    import ctypes as C
    so = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libzstd_ref_portable.so")
    if ref is None or not os.path.exists(so):
        pytest.skip("needs oracle/_ref")
    P = C.CDLL(os.path.abspath(so))
    _bind_zstd(oracle, P)
    rng = np.random.default_rng(4242)
    frames = [(f, d) for f, d in _direct_frames(ref) if 40 <= f.size <= 40000]
    pick = rng.permutation(len(frames))[:150]
    checked = accepted = 0
    for k in pick:
        frame, data = frames[int(k)]
        n = data.size
        for trial in range(16):
            c = _damage(rng, frame)
            ref_ok, orc_ok, _, want, got = _verdicts(oracle, P, c, n)
            assert ref_ok == orc_ok, (int(k), trial, frame.size, n)
            if ref_ok:
                assert np.array_equal(got, want), (int(k), trial)
                accepted += 1
            checked += 1
    assert checked >= 2400 and accepted >= 20, (checked, accepted)


def test_damaged_chunks_verdict_equal_reference(oracle, ref):
    """the same through the whole chunk path: blosc_decompress_ctx of the reference vs orc_decompress on damaged Zstd chunks"""
    from helpers import ref_decompress
    if ref is None:
        pytest.skip("needs oracle/_ref")
    _bind_zstd(oracle, ref)
    rng = np.random.default_rng(77)
    n = 1 << 18
    checked = stricter = 0
    for dname, T, clevel in (("bench19", 8, 3), ("smallints", 4, 5), ("linspace", 8, 1), ("randwalk", 8, 9)):
        data = DATASETS[dname](n)
        r, chunk = ref_compress(ref, data, T, clevel, 1, b"zstd")
        chunk = chunk[:r].copy()
        for trial in range(150):
            c = chunk.copy()
            pos = int(rng.integers(16, c.size))
            if trial % 3 == 2: c[pos] = int(rng.integers(0, 256))
            else: c[pos] ^= 1 << int(rng.integers(0, 8))
            rr, want = ref_decompress(ref, c, n)
            ro, got = orc_decompress(oracle, c, n)
            if rr != n:
                assert ro != n, (dname, trial, pos, rr, ro)
            elif ro == n:
                assert np.array_equal(got, want), (dname, trial, pos)
            else:
                oracle.orc_zstd_set_huf_lenient(1)
                try:
                    rl, _ = orc_decompress(oracle, c, n)
                finally:
                    oracle.orc_zstd_set_huf_lenient(0)
                assert rl == n, (dname, trial, pos)
                stricter += 1
            checked += 1
    assert checked == 600 and stricter * 5 <= checked, (checked, stricter)   # literal-heavy chunks: one flipped bit desynchronises a Huffman stream (63 of 600 with this seed)
