"""CPU, dev container only: the oracle equals the REAL reference (oracle/_ref/libblosc_ref.so, built from
the reference's own sources by oracle/Makefile) — filters, codec bytes, whole-chunk bytes, getitem."""
import numpy as np
import pytest

from _cmp_oracle_ref import compare_codecs, compare_filters
from helpers import DATASETS, orc_compress, ptr, ref_compress


def test_filters_and_codecs_bit_exact(oracle, ref):
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(0)
    assert compare_filters(ref, oracle, rng) == 0
    cnt, bad = compare_codecs(ref, oracle, rng, [0, 1, 12, 13, 16, 17, 66, 255, 4096, 65546, 65547, 131072])
    assert cnt > 3000 and bad == 0


def test_chunks_bit_exact(oracle, ref):
    if ref is None:
        pytest.skip("oracle/_ref not built")
    bad = []
    for cname in ["lz4", "blosclz"]:
        for T in [1, 4, 8, 17, 255]:
            for n in [0, 127, 128, 1000, 65536 + 17, 300001, (1 << 20) + 5]:
                for dname in ["bench19", "randwalk", "zeros", "random"]:
                    data = DATASETS[dname](n)
                    for clevel in [0, 1, 5, 9]:
                        for shuffle in [0, 1, 2]:
                            r1, c1 = ref_compress(ref, data, T, clevel, shuffle, cname.encode())
                            r2, c2 = orc_compress(oracle, data, T, clevel, shuffle, cname)
                            if r1 != r2 or (r1 > 0 and not np.array_equal(c1, c2)):
                                bad.append((cname, T, n, dname, clevel, shuffle, r1, r2))
                            if r1 > 0 and n >= T:
                                ni = n // T
                                a = np.zeros(ni * T + 1, np.uint8); b = np.zeros(ni * T + 1, np.uint8)
                                g1 = ref.blosc_getitem(ptr(c1), ni // 3, ni // 2, ptr(a)); g2 = oracle.orc_getitem(ptr(c1), ni // 3, ni // 2, ptr(b))
                                if g1 != g2 or not np.array_equal(a, b):
                                    bad.append(("getitem", cname, T, n, g1, g2))
    assert not bad, bad[:5]
