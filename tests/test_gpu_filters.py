"""GPU filter kernels == oracle (== blosc_internal_* of the reference), byte for byte.
Grid follows the reference's tests/test_shuffle_roundtrip_*.csv (T x N) plus bit-shuffle rules."""
import numpy as np
import pytest

from helpers import ptr

pytestmark = pytest.mark.gpu

TYPESIZES = [1, 2, 3, 4, 5, 6, 7, 8, 11, 16, 22, 30, 32, 42, 48, 52, 53, 64, 80, 255]
NELEMS = [7, 192, 500, 1792, 8000, 100000, 702713 // 8]


@pytest.mark.parametrize("T", TYPESIZES)
def test_shuffle_unshuffle_vs_oracle(lib, oracle, T):
    rng = np.random.default_rng(T)
    for N in NELEMS:
        for tail in (0, T // 2):
            n = N * T + tail
            src = rng.integers(0, 256, n, dtype=np.uint8)
            want = np.zeros(n, np.uint8); got = np.full(n, 0xA5, np.uint8)
            oracle.orc_shuffle(T, n, ptr(src), ptr(want))
            lib.blosc_internal_shuffle(T, n, ptr(src), ptr(got))
            assert np.array_equal(want, got), f"shuffle T={T} n={n}"
            back = np.full(n, 0x5A, np.uint8)
            lib.blosc_internal_unshuffle(T, n, ptr(got), ptr(back))
            assert np.array_equal(back, src), f"unshuffle T={T} n={n}"


@pytest.mark.parametrize("T", TYPESIZES)
def test_bitshuffle_vs_oracle(lib, oracle, T):
    rng = np.random.default_rng(100 + T)
    for N in [8, 64, 192, 1000, 1792, 8000, 100000, 7, 501]:   # 7 and 501: not multiples of 8 -> verbatim copy
        for tail in (0, T // 2):
            n = N * T + tail
            src = rng.integers(0, 256, n, dtype=np.uint8)
            want = np.zeros(n, np.uint8); got = np.full(n, 0xA5, np.uint8)
            r0 = oracle.orc_bitshuffle(T, n, ptr(src), ptr(want))
            r1 = lib.blosc_internal_bitshuffle(T, n, ptr(src), ptr(got), None)
            assert r0 == r1 and np.array_equal(want, got), f"bitshuffle T={T} n={n}"
            back = np.full(n, 0x5A, np.uint8)
            lib.blosc_internal_bitunshuffle(T, n, ptr(got), ptr(back), None)
            assert np.array_equal(back, src), f"bitunshuffle T={T} n={n}"


def test_filters_full_block_sizes(lib, oracle):
    """Config geometry: 1 MiB block T=8 (cfg #2) and 512 KiB block T=4 bitshuffle (cfg #3)."""
    rng = np.random.default_rng(5)
    for T, n, bit in [(8, 1 << 20, False), (4, 1 << 19, True), (16, 1 << 20, False), (2, 1 << 16, False), (8, 1 << 20, True)]:
        src = rng.integers(0, 256, n, dtype=np.uint8)
        want = np.zeros(n, np.uint8); got = np.zeros(n, np.uint8)
        if bit:
            oracle.orc_bitshuffle(T, n, ptr(src), ptr(want)); lib.blosc_internal_bitshuffle(T, n, ptr(src), ptr(got), None)
        else:
            oracle.orc_shuffle(T, n, ptr(src), ptr(want)); lib.blosc_internal_shuffle(T, n, ptr(src), ptr(got))
        assert np.array_equal(want, got)
