"""CPU: oracle/zstd_oracle.c (Zstd frame decoder restated from the format) pinned against the reference:
the compat/*zstd*.cdata golden vectors, committed chunks written by the real reference
(tests/golden/ref_zstd_chunks.npz, made by make_ref_zstd_chunks.py) and, where oracle/_ref exists, a sweep of
freshly written chunks incl. getitem.  Groundwork for codec row K8: the GPU product still answers
-5 for Zstd chunks."""
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
