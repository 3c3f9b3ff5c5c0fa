"""Data sets and small wrappers shared by the tests (SURVEY.md §8d inputs)."""
import ctypes as C

import numpy as np

CODECS = {"blosclz": 0, "lz4": 1}


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def bench19(nbytes):
    """bench/bench.c:141-170 generator: int32 v[i] = ((i<<26)^(i<<18)^(i<<11)^(i<<3)^i) & (2^19-1)."""
    i = np.arange((nbytes + 3) // 4, dtype=np.int64)
    v = ((i << 26) ^ (i << 18) ^ (i << 11) ^ (i << 3) ^ i) & ((1 << 19) - 1)
    return v.astype("<i4").view(np.uint8)[:nbytes].copy()


def linspace_f64(nbytes):
    n = (nbytes + 7) // 8
    return np.linspace(0, 1, n).astype("<f8").view(np.uint8)[:nbytes].copy()


def randwalk_f64(nbytes, seed=42):
    n = (nbytes + 7) // 8
    return np.cumsum(np.random.default_rng(seed).standard_normal(n)).astype("<f8").view(np.uint8)[:nbytes].copy()


def randbytes(nbytes, seed=1234):
    return np.frombuffer(np.random.default_rng(seed).bytes(nbytes), np.uint8).copy()


def arange_i32(nbytes):
    return np.arange((nbytes + 3) // 4, dtype="<i4").view(np.uint8)[:nbytes].copy()


def smallints_i32(nbytes, seed=7):
    return np.random.default_rng(seed).integers(0, 1 << 12, (nbytes + 3) // 4).astype("<i4").view(np.uint8)[:nbytes].copy()


def synth_segments(nbytes, seed=99):
    """Not a SURVEY data set: a patchwork for bug hunting (tests/test_gpu_parity_sweep.py, scripts/parity_hunt.py).  Segments of 1 byte ... 300 KiB laid end
    to end: zeros, one repeated byte, short periods (1 ... 300 bytes, powers of two and not), noise, noise over a few values, counters of 1 / 2 / 4 / 8 byte
    width, slowly varying float64, and COPIES of earlier ranges at distances from a few bytes to beyond 64 KiB (beyond any LZ window), with segment borders that
    land on, right before and right behind 1 KiB / 64 KiB / block boundaries."""
    rng = np.random.default_rng(seed)
    out = np.empty(nbytes, np.uint8)
    p = 0
    while p < nbytes:
        u = rng.random()
        n = int(rng.choice([1, 3, 17, 255, 1024, 4096, 65536])) if u < 0.3 else int(rng.integers(1, 300 << 10))
        if rng.random() < 0.15:                              # end exactly on / around a boundary
            b = int(rng.choice([1024, 65536, 131072, 1 << 20]))
            n = (-p) % b + int(rng.choice([0, 0, 1, b - 1, -1 % b]))
            n = max(n, 1)
        n = min(n, nbytes - p)
        kind = int(rng.integers(0, 9))
        if kind == 0: seg = np.zeros(n, np.uint8)
        elif kind == 1: seg = np.full(n, int(rng.integers(0, 256)), np.uint8)
        elif kind == 2:
            per = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 31, 32, 64, 100, 128, 256, 300, 1024, 4096]))
            seg = np.resize(rng.integers(0, 256, per, dtype=np.uint8), n)
        elif kind == 3: seg = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 4: seg = rng.integers(0, int(rng.choice([2, 3, 16])), n, dtype=np.uint8)
        elif kind == 5:
            w = int(rng.choice([1, 2, 4, 8]))
            cnt = (int(rng.integers(0, 1 << 20)) + np.arange((n + w - 1) // w, dtype=np.uint64) * int(rng.choice([1, 1, 3, 256])))
            seg = cnt.astype({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[w]).view(np.uint8)[:n]
        elif kind == 6:
            x = np.cumsum(rng.standard_normal((n + 7) // 8) * 1e-3) + float(rng.integers(-5, 5))
            seg = x.view(np.uint8)[:n]
        elif p > 0:
            d = int(rng.choice([1, 2, 8, 100, 1024, 8192, 65535, 65536, 70000, 200000])) if rng.random() < 0.6 else int(rng.integers(1, p + 1))
            d = min(d, p)
            seg = np.empty(n, np.uint8)
            for k in range(0, n, d):                          # (a copy that may overlap itself, like an LZ match)
                m = min(d, n - k); seg[k:k + m] = out[p - d:p - d + m] if k == 0 else seg[k - d:k - d + m]
        else:
            seg = np.zeros(n, np.uint8)
        out[p:p + n] = seg
        p += n
    return out


DATASETS = {
    "bench19": bench19, "linspace": linspace_f64, "randwalk": randwalk_f64, "random": randbytes,
    "arange": arange_i32, "smallints": smallints_i32,
    "zeros": lambda n: np.zeros(n, np.uint8),
    "synth": synth_segments,
}


def orc_compress(O, data, typesize, clevel=5, shuffle=1, codec="lz4", blocksize=0, destsize=None, splitmode=4):
    data = np.ascontiguousarray(data)
    cap = data.size + 16 if destsize is None else destsize
    out = np.zeros(max(cap, 16) + 64, np.uint8)
    r = O.orc_compress(clevel, shuffle, typesize, data.size, ptr(data), ptr(out), cap, CODECS[codec], blocksize, splitmode)
    return r, (out[:r].copy() if r > 0 else None)


def orc_decompress(O, chunk, nbytes):
    chunk = np.ascontiguousarray(chunk)
    out = np.zeros(max(nbytes, 1), np.uint8)
    r = O.orc_decompress(ptr(chunk), ptr(out), nbytes)
    return r, out[:max(r, 0)]


def ref_compress(R, data, typesize, clevel=5, shuffle=1, codec=b"lz4", blocksize=0, nthreads=1):
    data = np.ascontiguousarray(data)
    out = np.zeros(data.size + 16 + 64, np.uint8)
    r = R.blosc_compress_ctx(clevel, shuffle, typesize, data.size, ptr(data), ptr(out), data.size + 16, codec, blocksize, nthreads)
    return r, (out[:r].copy() if r > 0 else None)


def ref_decompress(R, chunk, nbytes):
    chunk = np.ascontiguousarray(chunk)
    out = np.zeros(max(nbytes, 1), np.uint8)
    r = R.blosc_decompress_ctx(ptr(chunk), ptr(out), nbytes, 1)
    return r, out[:max(r, 0)]


def header(chunk):
    c = np.asarray(chunk[:16], np.uint8)
    return dict(version=int(c[0]), versionlz=int(c[1]), flags=int(c[2]), typesize=int(c[3]),
                nbytes=int(c[4:8].view("<i4")[0]), blocksize=int(c[8:12].view("<i4")[0]), cbytes=int(c[12:16].view("<i4")[0]))


def wrap_stream_as_chunk(stream, nbytes, fmt):
    """A one-block, unsplit chunk around ONE codec stream (for hand-built LZ streams).
    fmt: 0 BloscLZ, 1 LZ4.  typesize 1, no filter, dont_split set."""
    stream = np.asarray(stream, np.uint8)
    total = 16 + 4 + 4 + stream.size
    c = np.zeros(total, np.uint8)
    c[0] = 2; c[1] = 1; c[2] = 0x10 | (fmt << 5); c[3] = 1
    c[4:8] = np.array([nbytes], "<i4").view(np.uint8)
    c[8:12] = np.array([nbytes], "<i4").view(np.uint8)
    c[12:16] = np.array([total], "<i4").view(np.uint8)
    c[16:20] = np.array([20], "<i4").view(np.uint8)
    c[20:24] = np.array([stream.size], "<i4").view(np.uint8)
    c[24:] = stream
    return c


def wrap_planes_as_chunk(streams, neblock, fmt, shuffle_flag=1):
    """A one-block chunk whose block is SPLIT into len(streams) codec streams (typesize = len(streams),
    byte-shuffle flag set): what blosc_c writes for a shuffled block (blosc/blosc.c:635-719), with every
    stream hand-built.  Each stream must decode to exactly `neblock` bytes."""
    T = len(streams)
    nbytes = T * neblock
    body = bytearray()
    for s in streams:
        s = bytes(s)
        body += np.array([len(s)], "<i4").tobytes() + s
    total = 16 + 4 + len(body)
    c = np.zeros(total, np.uint8)
    c[0] = 2; c[1] = 1; c[2] = shuffle_flag | (fmt << 5); c[3] = T
    c[4:8] = np.array([nbytes], "<i4").view(np.uint8)
    c[8:12] = np.array([nbytes], "<i4").view(np.uint8)
    c[12:16] = np.array([total], "<i4").view(np.uint8)
    c[16:20] = np.array([20], "<i4").view(np.uint8)
    c[20:] = np.frombuffer(bytes(body), np.uint8)
    return c
