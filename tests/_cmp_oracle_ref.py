"""Ad-hoc driver (also imported by tests/test_oracle_vs_reference.py): compares the oracle's
filters and codecs with the real reference (oracle/_ref/libblosc_ref.so)."""
import ctypes as C
import numpy as np

vp = C.c_void_p


def ptr(a):
    return a.ctypes.data_as(vp)


def datasets(n, rng):
    i = np.arange(n // 4 + 1, dtype=np.int64)
    b19 = (((i << 26) ^ (i << 18) ^ (i << 11) ^ (i << 3) ^ i) & ((1 << 19) - 1)).astype('<i4').view(np.uint8)[:n]
    yield 'bench19', b19
    yield 'zeros', np.zeros(n, np.uint8)
    yield 'rand', rng.integers(0, 256, n, dtype=np.uint8)
    yield 'rand4', rng.integers(0, 4, n, dtype=np.uint8)
    yield 'ramp', (np.arange(n) % 251).astype(np.uint8)
    yield 'walk', np.cumsum(rng.standard_normal(n // 8 + 1)).view(np.uint8)[:n]
    sh = np.ascontiguousarray(b19[:n // 8 * 8].reshape(-1, 8).T).ravel()
    yield 'b19shuf', np.concatenate([sh, b19[len(sh):]])[:n]


def compare_filters(R, O, rng):
    bad = 0
    for T in [1, 2, 3, 4, 5, 7, 8, 11, 16, 17, 32, 33, 255]:
        for n in [0, 1, 7, 15, 16, 127, 128, 192, 500, 1792, 8000, 100003]:
            src = rng.integers(0, 256, n, dtype=np.uint8)
            a = np.zeros(n, np.uint8); b = np.zeros(n, np.uint8); t = np.zeros(n + 64, np.uint8)
            cT, cn = C.c_size_t(T), C.c_size_t(n)
            R.blosc_internal_shuffle(cT, cn, ptr(src), ptr(a)); O.orc_shuffle(cT, cn, ptr(src), ptr(b))
            bad += not np.array_equal(a, b)
            R.blosc_internal_unshuffle(cT, cn, ptr(src), ptr(a)); O.orc_unshuffle(cT, cn, ptr(src), ptr(b))
            bad += not np.array_equal(a, b)
            if n >= T:
                r1 = R.blosc_internal_bitshuffle(cT, cn, ptr(src), ptr(a), ptr(t)); r2 = O.orc_bitshuffle(cT, cn, ptr(src), ptr(b))
                bad += (not np.array_equal(a, b)) or r1 != r2
                r1 = R.blosc_internal_bitunshuffle(cT, cn, ptr(src), ptr(a), ptr(t)); r2 = O.orc_bitunshuffle(cT, cn, ptr(src), ptr(b))
                bad += (not np.array_equal(a, b)) or r1 != r2
    return bad


def compare_codecs(R, O, rng, sizes, verbose=False):
    bad = 0; cnt = 0
    for n in sizes:
        for name, d in datasets(n, rng):
            d = np.ascontiguousarray(d)
            for cap in sorted(set([max(n, 0), n + n // 255 + 16, max(n - 1, 0), n // 2, 66, 10])):
                for acc in [1, 5, 9]:
                    a = np.zeros(cap + 8, np.uint8); b = np.zeros(cap + 8, np.uint8)
                    r1 = R.LZ4_compress_fast(ptr(d), ptr(a), n, cap, acc); r2 = O.orc_lz4_compress(ptr(d), n, ptr(b), cap, acc); cnt += 1
                    if r1 != r2 or not np.array_equal(a[:max(r1, 0)], b[:max(r2, 0)]):
                        bad += 1
                        if verbose: print('LZ4 MISMATCH', n, name, cap, acc, r1, r2)
                    if r1 > 0:
                        o1 = np.zeros(n + 1, np.uint8); o2 = np.zeros(n + 1, np.uint8)
                        q1 = R.LZ4_decompress_safe(ptr(a), ptr(o1), r1, n); q2 = O.orc_lz4_decompress(ptr(a), r1, ptr(o2), n)
                        if q1 != q2 or q1 != n or not np.array_equal(o2[:n], d[:n]):
                            bad += 1
                            if verbose: print('LZ4 DEC MISMATCH', n, name, q1, q2)
                for cl in [1, 3, 5, 9]:
                    for sp in [0, 1]:
                        a = np.zeros(cap + 8, np.uint8); b = np.zeros(cap + 8, np.uint8)
                        r1 = R.blosclz_compress(cl, ptr(d), n, ptr(a), cap, sp); r2 = O.orc_blosclz_compress(cl, ptr(d), n, ptr(b), cap, sp); cnt += 1
                        if r1 != r2 or not np.array_equal(a[:max(r1, 0)], b[:max(r2, 0)]):
                            bad += 1
                            if verbose: print('BLZ MISMATCH', n, name, cap, cl, sp, r1, r2)
                        if r1 > 0:
                            o1 = np.zeros(n + 1, np.uint8); o2 = np.zeros(n + 1, np.uint8)
                            q1 = R.blosclz_decompress(ptr(a), r1, ptr(o1), n); q2 = O.orc_blosclz_decompress(ptr(a), r1, ptr(o2), n)
                            if q1 != q2 or q1 != n or not np.array_equal(o2[:n], d[:n]):
                                bad += 1
                                if verbose: print('BLZ DEC MISMATCH', n, name, q1, q2)
    return cnt, bad


if __name__ == '__main__':
    R = C.CDLL('/root/repo/oracle/_ref/libblosc_ref.so'); O = C.CDLL('/root/repo/oracle/liboracle.so')
    rng = np.random.default_rng(0)
    print('filter mismatches', compare_filters(R, O, rng))
    print('codec (cases, mismatches)', compare_codecs(R, O, rng,
          [0, 1, 5, 12, 13, 14, 15, 16, 17, 31, 64, 65, 66, 100, 255, 1000, 4096, 65546, 65547, 65548, 131072, 300000], True))
