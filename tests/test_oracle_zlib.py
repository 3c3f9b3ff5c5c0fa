"""CPU: oracle/zlib_oracle.c (zlib stream decoder restated from RFC 1950 / 1951) pinned against the reference: the five
compat/*zlib*.cdata golden vectors, committed chunks written by the real reference (tests/golden/ref_zlib_chunks.npz, made by
make_ref_zlib_chunks.py), hand-built legal and illegal streams (tests/deflate_builder.py) and - where oracle/_ref exists -
the reference's own `uncompress` as the live yardstick for verdict AND bytes on valid, truncated and corrupted streams.
The GPU decoder (k_zlib.hip) is checked against this oracle, the reference and the same fixtures in tests/test_gpu_zlib.py."""
import ctypes as C
import glob
import os
import zlib

import numpy as np
import pytest

import deflate_builder as D
from helpers import DATASETS, orc_decompress, ptr, ref_compress

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
# verdicts of the reference's zlib for tests/deflate_builder.py: cases() (name -> bytes produced, 0 = rejected), recorded with
# oracle/_ref and re-checked live by test_handbuilt_streams_against_reference whenever the reference is there
HANDBUILT = {"dist-one-symbol": 9, "no-dist-code": 3, "eob-only-then-stored": 3, "lit-incomplete": 0, "lit-oversubscribed": 0,
             "no-eob-code": 0, "cl-incomplete": 0, "repeat-at-start": 0, "zero-run-across-boundary": 2, "repeat-past-end": 0,
             "fixed-legal": 277, "fixed-sym-286": 0, "fixed-dist-30": 0, "dist-too-far": 0, "stored-bad-nlen": 0,
             "fixed-stored-fixed": 10, "bad-adler": 0, "bad-fcheck": 0, "fdict": 0, "cinfo-0": 1, "cm-7": 0, "dist-two-symbols": 9,
             "dist-one-symbol-eob-only": 0, "hlit-287": 0, "fixed-no-eob": 0, "fixed-empty": 0, "btype-3": 0}


def _zo(oracle):
    oracle.orc_zlib_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return oracle.orc_zlib_decompress


def _un(fn, s, cap):
    src = np.empty(s.size, np.uint8); src[:] = s          # exact-size copy: an overrun is one byte past an allocation
    out = np.zeros(max(cap, 1), np.uint8)
    return fn(ptr(src), src.size, ptr(out), cap), out


def ref_uncompress(ref):
    ref.uncompress.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.c_ulong]
    def fn(p, n, o, cap):
        ul = C.c_ulong(cap)
        return int(ul.value) if ref.uncompress(o, C.byref(ul), p, n) == 0 else 0       # zlib_wrap_decompress, blosc.c:484-495
    return fn


def stock_streams(sizes=(1, 3, 17, 255, 4096, 70000, 131072)):
    """streams written by zlib itself (python's, same format) at every level, with the fixed / Huffman-only / RLE strategies,
    small windows and full flushes (several blocks, stored blocks at level 0)"""
    for name in ("bench19", "linspace", "randwalk", "random", "zeros", "smallints"):
        for n in sizes:
            d = DATASETS[name](n); b = d.tobytes()
            for lvl in (0, 1, 6, 9):
                yield np.frombuffer(zlib.compress(b, lvl), np.uint8).copy(), d
            for strat, wb in ((zlib.Z_FIXED, 15), (zlib.Z_HUFFMAN_ONLY, 15), (zlib.Z_RLE, 9)):
                co = zlib.compressobj(6, zlib.DEFLATED, wb, 8, strat)
                yield np.frombuffer(co.compress(b) + co.flush(), np.uint8).copy(), d
            co = zlib.compressobj(5); step = max(1, len(b) // 5); parts = []
            for k in range(0, len(b), step):
                parts += [co.compress(b[k:k + step]), co.flush(zlib.Z_FULL_FLUSH)]
            yield np.frombuffer(b"".join(parts) + co.flush(), np.uint8).copy(), d


def mutate(s, t, rng):
    c = s.copy(); m = t % 6
    if m == 0 and c.size > 1: c = c[:int(rng.integers(1, c.size))].copy()
    elif m == 1: c = np.concatenate([c, rng.integers(0, 256, int(rng.integers(1, 9))).astype(np.uint8)])
    elif m == 2: c[int(rng.integers(0, c.size))] = int(rng.integers(0, 256))
    else: c[int(rng.integers(0, min(c.size, 64) if m == 3 else c.size))] ^= 1 << int(rng.integers(0, 8))
    return c


@pytest.mark.parametrize("fname", sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "compat", "*zlib*.cdata"))))
def test_compat_zlib_vectors(oracle, fname):
    chunk = np.fromfile(os.path.join(GOLDEN, "compat", fname), np.uint8)
    r, out = orc_decompress(oracle, chunk, 4000000)
    assert r == 4000000 and np.array_equal(out.view("<i4"), np.arange(10**6, dtype="<i4"))


def test_committed_reference_chunks(oracle):
    z = np.load(os.path.join(GOLDEN, "ref_zlib_chunks.npz"))
    meta = [m.split(",") for m in z["meta"]]
    assert len(meta) >= 24
    for k, (dname, n, T, clevel, shuffle, bs) in enumerate(meta):
        n = int(n)
        r, out = orc_decompress(oracle, z[f"c{k}"], n)
        assert r == n and np.array_equal(out, DATASETS[dname](n)), meta[k]


def test_stock_streams(oracle):
    zo = _zo(oracle); cnt = 0
    for s, d in stock_streams():
        r, out = _un(zo, s, d.size)
        assert r == d.size and np.array_equal(out[:r], d)
        assert _un(zo, s, d.size - 1)[0] == 0 if d.size > 1 else True       # output does not fit: failure
        r2, out2 = _un(zo, s, d.size + 7)                                    # room to spare: the true size comes back
        assert r2 == d.size and np.array_equal(out2[:r2], d)
        cnt += 1
    assert cnt >= 300


def test_handbuilt_streams(oracle):
    zo = _zo(oracle)
    cases = D.cases()
    assert {c[0] for c in cases} == set(HANDBUILT)
    for name, s, cap in cases:
        assert _un(zo, s, max(cap, 1) + 300)[0] == HANDBUILT[name], name


def test_handbuilt_streams_against_reference(ref):
    if ref is None:
        pytest.skip("oracle/_ref not built here")
    ru = ref_uncompress(ref)
    for name, s, cap in D.cases():
        assert _un(ru, s, max(cap, 1) + 300)[0] == HANDBUILT[name], name


def test_corrupt_streams_against_reference(oracle, ref):
    """verdict and bytes equal the reference's on truncated / extended / bit-flipped streams"""
    if ref is None:
        pytest.skip("oracle/_ref not built here")
    zo = _zo(oracle); ru = ref_uncompress(ref)
    rng = np.random.default_rng(5); ntot = nacc = 0
    for s, d in stock_streams(sizes=(3, 255, 4096, 70000)):
        for t in range(12 if s.size < 20000 else 4):
            c = mutate(s, t, rng)
            r, out = _un(zo, c, d.size); rr, oo = _un(ru, c, d.size)
            assert r == rr and np.array_equal(out[:r], oo[:rr]), (t, s.size, r, rr)
            ntot += 1; nacc += rr > 0
    assert ntot > 1500 and nacc > 100


def test_truncated_and_corrupt_chunks_fail_cleanly(oracle):
    z = np.load(os.path.join(GOLDEN, "ref_zlib_chunks.npz"))
    chunk = z["c0"]; n = int(z["meta"][0].split(",")[1])
    rng = np.random.default_rng(3)
    for trial in range(300):
        c = chunk.copy()
        pos = int(rng.integers(16, c.size)); c[pos] ^= 1 << int(rng.integers(0, 8))
        out = np.zeros(n, np.uint8)
        r = oracle.orc_decompress(ptr(c), ptr(out), n)
        assert r == n or r < 0


def test_fresh_reference_chunks(oracle, ref):
    """sweep against the real reference (only where oracle/_ref is built), incl. getitem"""
    if ref is None:
        pytest.skip("oracle/_ref not built here")
    for dname in ("bench19", "linspace", "randwalk", "smallints"):
        for clevel in (1, 5, 9):
            for T, shuffle in ((8, 1), (4, 2), (2, 1), (1, 0), (32, 1)):
                n = 1 << 19
                data = DATASETS[dname](n)
                r, chunk = ref_compress(ref, data, T, clevel, shuffle, b"zlib")
                assert r > 0
                d, out = orc_decompress(oracle, chunk, n)
                assert d == n and np.array_equal(out, data), (dname, clevel, T, shuffle)
                start, nitems = 1000, 5000
                got = np.zeros(nitems * T, np.uint8); want = np.zeros(nitems * T, np.uint8)
                assert oracle.orc_getitem(ptr(chunk), start, nitems, ptr(got)) == nitems * T
                assert ref.blosc_getitem(ptr(chunk), start, nitems, ptr(want)) == nitems * T
                assert np.array_equal(got, want)
