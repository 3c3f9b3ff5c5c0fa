"""CPU: the oracle against the reference's golden vectors (compat/*.cdata, committed under
tests/golden/compat) and against chunks written by the real reference (tests/golden/ref_chunks.npz)."""
import glob
import os

import numpy as np
import pytest

from helpers import DATASETS, orc_compress, orc_decompress, ptr

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("fname", sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "compat", "*.cdata"))))
def test_compat_vectors(oracle, fname):
    chunk = np.fromfile(os.path.join(GOLDEN, "compat", fname), np.uint8)
    r, out = orc_decompress(oracle, chunk, 4000000)
    if "snappy" in fname:
        assert r == -5          # codec outside this build's scope: same code a stock build without it gives
    else:
        assert r == 4000000 and np.array_equal(out.view("<i4"), np.arange(10**6, dtype="<i4"))


def test_reference_written_chunks(oracle):
    z = np.load(os.path.join(GOLDEN, "ref_chunks.npz"))
    meta = [m.split(",") for m in z["meta"]]
    assert len(meta) > 100
    for k, (cname, shuffle, dname, n, T, clevel, nth, bs, sm) in enumerate(meta):
        n = int(n)
        data = DATASETS[dname](n)
        r, out = orc_decompress(oracle, z[f"c{k}"], n)
        assert r == n and np.array_equal(out, data), meta[k]
        # single-threaded lz4 / blosclz chunks are also byte-identical to what the oracle writes
        if cname in ("lz4", "blosclz") and nth == "1":
            r2, mine = orc_compress(oracle, data, int(T), int(clevel), int(shuffle), cname, int(bs), splitmode=int(sm))
            assert r2 == z[f"c{k}"].size and np.array_equal(mine, z[f"c{k}"]), meta[k]


def test_roundtrip_and_sizes(oracle):
    """tests/test_maxout.c / test_compressor.c known answers on the oracle."""
    data = DATASETS["random"](1000)
    out = np.zeros(2000, np.uint8)
    C = oracle.orc_compress
    assert C(5, 1, 1, 1000, ptr(data), ptr(out), 1015, 1, 0, 4) == 0
    assert C(5, 1, 1, 1000, ptr(data), ptr(out), 1016, 1, 0, 4) == 1016
    assert C(5, 1, 1, 1000, ptr(data), ptr(out), 15, 1, 0, 4) == 0
    assert C(5, 1, 4, 0, ptr(data), ptr(out), 2000, 0, 0, 4) == 16
    for n in range(1, 16):
        assert C(5, 1, 4, n, ptr(data), ptr(out), 2000, 0, 0, 4) == n + 16
    assert C(10, 1, 4, 1000, ptr(data), ptr(out), 2000, 1, 0, 4) == -10
    assert C(5, 3, 4, 1000, ptr(data), ptr(out), 2000, 1, 0, 4) == -10
    assert C(5, 1, 0, 1000, ptr(data), ptr(out), 2000, 1, 0, 4) == -10
