"""CPU: the format-writing half of the GPU Zstd encoder (c-blosc_amd/csrc/zstd_enc.h: frame / block / raw-literals
headers, sequence codes, predefined FSE tables as the encoder sees them, per-block tables - normalised counts, their
FSE table description, RLE mode -, Huffman-coded literals (complete length-limited codes, tree description direct or with
FSE-compressed weights on two interleaved states, one or four streams), repeat-offset codes, the backward bitstream) compiled with g++ behind a plain greedy matcher (tests/tools/zstd_enc_cpu.cpp).  Every frame must be read back
bit-exactly by the oracle's decoder and - where oracle/_ref ships - by the reference's own ZSTD_decompress."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import DATASETS, ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def enc():
    so = os.path.join(ROOT, "tests", "tools", "libzstd_enc_cpu.so")
    src = os.path.join(ROOT, "tests", "tools", "zstd_enc_cpu.cpp")
    hdr = os.path.join(ROOT, "c-blosc_amd", "csrc", "zstd_enc.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, src])
    E = C.CDLL(so)
    E.zenc_cpu_compress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    E.zenc_cpu_compress2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    return E


def test_frames_decode_with_oracle_and_reference(enc, oracle, ref):
    oracle.orc_zstd_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    if ref is not None:
        ref.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        ref.ZSTD_decompress.restype = C.c_size_t
        ref.ZSTD_isError.argtypes = [C.c_size_t]
    cases = 0
    sizes = {}
    for dname in ["bench19", "linspace", "randwalk", "smallints", "zeros", "random"]:
        for n in [1, 5, 100, 1000, 4096, 65536, 131072, 131073, 300001, 1 << 20]:      # 131073+: several blocks per frame
            data = DATASETS[dname](n)
            if dname != "random" and n >= 4096 and n % 8 == 0:
                data = data.reshape(-1, 8).T.copy().reshape(-1)                          # byte planes, as inside a blosc block
            for minmatch, tables in ((3, 0), (4, 0), (8, 0), (3, 1), (4, 1), (8, 1), (4, 2), (4, 3), (8, 3)):
                out = np.zeros(n + n // 8 + 64, np.uint8)
                r = enc.zenc_cpu_compress2(ptr(data), n, ptr(out), out.size, minmatch, tables, None, 0)
                assert r > 0
                sizes[tables] = sizes.get(tables, 0) + r
                back = np.zeros(n + 8, np.uint8)
                assert oracle.orc_zstd_decompress(ptr(out), r, ptr(back), n) == n and np.array_equal(back[:n], data), (dname, n, minmatch)
                if ref is not None:
                    back2 = np.zeros(n + 8, np.uint8)
                    d = ref.ZSTD_decompress(ptr(back2), n, ptr(out), r)
                    assert not ref.ZSTD_isError(d) and d == n and np.array_equal(back2[:n], data), (dname, n, minmatch)
                cases += 1
    assert cases == 540
    assert sizes[1] < sizes[0] and sizes[3] < sizes[1]                       # per-block tables are only taken where they pay


def test_too_small_destination(enc):
    data = DATASETS["random"](5000)
    out = np.zeros(8000, np.uint8)
    assert enc.zenc_cpu_compress(ptr(data), 5000, ptr(out), 4000, 4) == 0        # does not fit: the caller stores the split raw


def test_table_descriptions_cover_the_corner_cases(enc, ref):
    """FSE table descriptions (RFC 8878 4.1.1) with long runs of absent symbols (the 2-bit repeat flags, chained), one dominant
    symbol, as many symbols as cells, and the RLE case - all through hand-made sequence lists, read by ZSTD_decompress."""
    if ref is None:
        pytest.skip("needs the reference's ZSTD_decompress (oracle/_ref)")
    ref.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    ref.ZSTD_decompress.restype = C.c_size_t
    rng = np.random.default_rng(11)
    tried = 0
    for trial in range(300):
        # a block described by (literal length, match length, distance) triples with deliberately odd code statistics
        kind = trial % 6
        nseq = int(rng.integers(2, 400))
        if kind == 0:   ll = np.zeros(nseq, np.int64); ml = np.full(nseq, 4); off = np.full(nseq, 1)                          # every alphabet RLE
        elif kind == 1: ll = rng.choice([0, 70000 // nseq], nseq); ml = rng.choice([3, 200], nseq, p=[0.97, 0.03]); off = rng.choice([1, 2], nseq)
        elif kind == 2: ll = rng.integers(0, 40, nseq); ml = rng.integers(3, 60, nseq); off = rng.integers(1, 300, nseq)      # many symbols
        elif kind == 3: ll = rng.choice([0, 1, 35], nseq, p=[0.9, 0.05, 0.05]); ml = rng.choice([3, 131], nseq); off = rng.choice([1, 1 << 14], nseq, p=[0.99, 0.01])
        elif kind == 4: ll = rng.integers(0, 3, nseq); ml = 3 + (1 << rng.integers(0, 9, nseq)); off = 1 << rng.integers(0, 10, nseq)
        else:           ll = rng.integers(0, 16, nseq); ml = np.full(nseq, 5); off = rng.integers(1, 4, nseq)
        ll = np.asarray(ll, np.int64); ml = np.asarray(ml, np.int64); off = np.asarray(off, np.int64)
        # build the data these sequences describe
        out = bytearray(rng.integers(0, 256, 1 << 14, dtype=np.uint8).tobytes())                  # history the first matches reach into
        base = len(out)
        seqs = []
        for a, b, c in zip(ll, ml, off):
            c = int(min(c, len(out)))
            out += rng.integers(0, 256, int(a), dtype=np.uint8).tobytes()
            for _ in range(int(b)):
                out.append(out[-c])
            seqs.append((int(a), int(b), c))
            if len(out) - 0 > 120000:
                break
        data = np.frombuffer(bytes(out), np.uint8).copy()
        # the whole buffer is one block: the history part becomes one leading literal run
        seqs[0] = (seqs[0][0] + base, seqs[0][1], seqs[0][2])
        tri = np.array(seqs, np.uint32).reshape(-1)
        for tables in (0, 1):
            dst = np.zeros(data.size + 4096, np.uint8)
            r = enc.zenc_cpu_compress2(ptr(data), data.size, ptr(dst), dst.size, 4, tables, ptr(tri), len(seqs))
            assert r > 0, (trial, tables)
            back = np.zeros(data.size, np.uint8)
            d = ref.ZSTD_decompress(ptr(back), data.size, ptr(dst), r)
            assert d == data.size and np.array_equal(back, data), (trial, kind, tables, d)
            tried += 1
    assert tried == 600


def test_huffman_literals_on_odd_byte_distributions(enc, ref):
    """Literal alphabets of 2 .. 256 bytes, flat and extremely skewed (the 11-bit limit and its repair), values above 128 (weights
    must travel FSE-compressed), short and long runs (one stream / four streams, all three header sizes)."""
    if ref is None:
        pytest.skip("needs the reference's ZSTD_decompress (oracle/_ref)")
    ref.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    ref.ZSTD_decompress.restype = C.c_size_t
    rng = np.random.default_rng(1)
    smaller = 0
    for trial in range(240):
        n = int(rng.choice([40, 255, 256, 1000, 1023, 1024, 5000, 16383, 16384, 70000, 131072]))
        k = int(rng.choice([2, 3, 5, 16, 100, 129, 200, 256]))
        pr = np.random.default_rng(trial).dirichlet(np.ones(k) * rng.choice([0.02, 0.5, 5]))
        data = rng.choice(k, n, p=pr).astype(np.uint8)
        if trial % 3 == 0:
            data = (data.astype(np.int32) * int(rng.integers(1, 256 // k + 1))).astype(np.uint8)
        sizes = []
        for tables in (1, 3):
            out = np.zeros(n + 1024, np.uint8)
            r = enc.zenc_cpu_compress2(ptr(data), n, ptr(out), out.size, 4, tables, None, 0)
            assert r > 0
            back = np.zeros(n, np.uint8)
            d = ref.ZSTD_decompress(ptr(back), n, ptr(out), r)
            assert d == n and np.array_equal(back, data), (trial, n, k, tables)
            sizes.append(r)
        assert sizes[1] <= sizes[0]
        smaller += sizes[1] < sizes[0]
    assert smaller > 150
