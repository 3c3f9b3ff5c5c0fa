"""Periodic spans of the fused decode+unshuffle (c-blosc_amd/csrc/k_decode.hip: SpanCtx): planes whose
stream is "literals + one long match with a power-of-two distance" are not written to the scratch.
Hand-built LZ4 streams drive every branch: spans taken / refused, later matches that reach back into the
skipped range (materialise), sources at the span edges, a second long match, every alignment of the
match start.  The oracle decodes the same chunk; results must be bit-exact."""
import numpy as np
import pytest

from helpers import orc_decompress, ptr, wrap_planes_as_chunk
from test_gpu_decompress import _lz4_seq, _lz4_tail

pytestmark = pytest.mark.gpu

NEB = 128 << 10   # bytes per plane


def _plane_stream(rng, period, pre_extra, mlen, follow):
    """literals (period + pre_extra random bytes), a match of `mlen` at distance `period`, optional
    follow-up sequences, then the closing literals; decodes to exactly NEB bytes."""
    s = bytearray()
    produced = 0
    lit = rng.integers(0, 256, period + pre_extra, dtype=np.uint8).tobytes()
    s += _lz4_seq(lit, period, mlen); produced += len(lit) + mlen
    for (nlit, off, ml) in follow:
        off = min(off, produced + nlit, 65535)
        s += _lz4_seq(rng.integers(0, 256, nlit, dtype=np.uint8).tobytes(), off, ml); produced += nlit + ml
    tail = NEB - produced
    assert tail >= 12, (produced, tail)
    s += _lz4_tail(rng.integers(0, 256, tail, dtype=np.uint8).tobytes())
    return bytes(s)


def _check(pkg, oracle, streams):
    chunk = wrap_planes_as_chunk(streams, NEB, 1)
    n = len(streams) * NEB
    want = np.zeros(n, np.uint8)
    assert oracle.orc_decompress(ptr(chunk), ptr(want), n) == n
    r, out = pkg.decompress(chunk, n)
    assert r == n
    return np.array_equal(out, want)


FOLLOW = {
    "none": [],
    "into_span": [(7, 30000, 300)],                 # source far inside the skipped range
    "into_span_long": [(0, 50000, 20000)],
    "short_after": [(3, 5, 40), (0, 1, 30)],        # sources in the freshly written tail
    "edge_lo": [(5, 0, 64)],                        # patched below: source straddles `lo`
    "second_long": [(9, 256, 20000)],               # another span-able match while one span exists
    "many": [(1, 40000, 5), (2, 2049, 70), (0, 17000, 1024), (4, 64, 64)],
}


@pytest.mark.parametrize("period", [1, 2, 4, 8, 64, 256, 1024, 2048, 3, 100, 4096, 8192, 32768, 12288])
@pytest.mark.parametrize("follow", list(FOLLOW))
def test_span_streams(pkg, oracle, period, follow):
    rng = np.random.default_rng(period * 131 + len(follow))
    bad = []
    for pre_extra in (0, 1, 5, 1023, 1024, 1500):
        for mlen in (16383, 16384, 24577, 65536):
            fl = list(FOLLOW[follow])
            mpos = period + pre_extra
            if follow == "edge_lo":
                lo = (mpos + 1023) & ~1023
                end = mpos + mlen + 5
                fl = [(5, end - (lo - 20), 64)]     # source = [lo - 20, lo + 44)
            streams = [_plane_stream(rng, period, pre_extra, mlen, fl) for _ in range(2)]
            # the other planes: one plain noisy plane, constant planes, a short-period plane
            streams.append(_plane_stream(rng, 1, 0, NEB - 1 - 40, []))
            streams.append(_plane_stream(rng, 256, 3, 100000, [(2, 70000, 12)]))
            streams += [_plane_stream(rng, 7, 2, 500, [(100, 33, 900)]) for _ in range(2)]
            streams.append(_plane_stream(rng, 2048, 0, 120000, []))
            streams.append(_plane_stream(rng, 16, 1, 16400, [(0, 16, 100000)]))
            if not _check(pkg, oracle, streams):
                bad.append((pre_extra, mlen))
    assert not bad, bad


def test_span_typesize4(pkg, oracle):
    rng = np.random.default_rng(5)
    streams = [_plane_stream(rng, 1, 0, NEB - 100, []), _plane_stream(rng, 512, 9, 90000, [(3, 60000, 33)]),
               _plane_stream(rng, 5, 0, 20000, []), _plane_stream(rng, 4, 0, 131000, [])]
    assert _check(pkg, oracle, streams)


def test_self_span_base_beyond_24_bits_is_not_taken(pkg, oracle):
    """A periodic match with a period above the 2 KiB pattern table becomes a "self span" whose base travels in 24 bits next to log2(period)
    (k_decode.hip: unshuffle_block_wave_T).  A plane of 16 MiB or more can start such a match beyond that range - the reference never
    writes one (a split block is at most 1 MiB, blosc.c:1020-1040), a foreign writer may: a hand-built chunk of ONE 64 MiB block split into two
    32 MiB planes, the period-8192 match starting at plane position 16 MiB + 13 192.  It must go through the ring like any other match
    (dec_ring.h: dr_span_long_match refuses it); the oracle's reader says what the bytes are."""
    neb = 32 << 20
    rng = np.random.default_rng(24)
    head = (1 << 24) + 5000 + 8192
    lit = rng.integers(0, 256, head, dtype=np.uint8)
    s0 = _lz4_seq(lit.tobytes(), 8192, neb - head - 12) + _lz4_tail(rng.integers(0, 256, 12, dtype=np.uint8).tobytes())
    s1 = _lz4_seq(b"\x07", 1, neb - 1 - 12) + _lz4_tail(bytes(12))
    chunk = wrap_planes_as_chunk([s0, s1], neb, 1)
    n = 2 * neb
    ro, want = orc_decompress(oracle, chunk, n)
    assert ro == n
    got_r, got = pkg.decompress(chunk, n)
    assert got_r == n and np.array_equal(got, want)


@pytest.mark.parametrize("off2", [1, 2, 3, 5, 16, 33, 63])
def test_short_period_match_right_behind_a_span(pkg, oracle, off2):
    """Round 5: the reference's chunks of `linspace` float64 data compressed as typesize 4 decoded wrongly (silently: right sizes, wrong bytes) -
    their fourth byte plane is "2 literals, 32 766 bytes at distance 2, 1 literal, 32 767 bytes at distance 2, ...": the first match becomes a
    periodic span that ends exactly on a 1 KiB row boundary, so the ring holds nothing below it, and the second match - a SHORT period (< 64),
    longer than its distance - took its period bytes from the ring all the same (dec_ring.h: dr_match).  Hand-built: a span that ends on a row
    boundary / one, two, 900 bytes behind one, followed by 0 .. 3 literals and a long match of distance 1 .. 63 whose source reaches below the span's
    end; twice in a row; in every plane position of a typesize-4 and a typesize-8 block.  The oracle says what the bytes are."""
    rng = np.random.default_rng(900 + off2)
    bad = []
    for period in (2, 1, 64, 1024):
        for end_mis in (0, 1, 2, 900):                        # where the span's match ends relative to a row boundary
            for nlit in (0, 1, 3):
                if nlit + end_mis < 1 and off2 > period and period != 1:
                    pass                                        # (a source that straddles the span's last period: fine, the oracle defines it)
                pre = 7
                mlen = 32768 + end_mis - (period + pre)        # first match ends at 32 KiB + end_mis
                follow = [(nlit, off2, 20000), (1, min(off2, 2), 30000)]
                a = _plane_stream(rng, period, pre, mlen, follow)
                for T in (4, 8):
                    planes = [_plane_stream(rng, 1, 0, NEB - 1 - 40, []) for _ in range(T)]
                    for pos in range(T):
                        streams = list(planes); streams[pos] = a
                        if not _check(pkg, oracle, streams):
                            bad.append((period, end_mis, nlit, T, pos))
    assert not bad, bad[:10]


def test_reference_linspace_chunk_at_typesize_4(pkg, oracle):
    """... and the data that showed it: 16 MiB of linspace float64 labelled typesize 4, written by the reference's algorithm (the oracle's writer is
    byte-identical to it), every byte compared."""
    from helpers import DATASETS, orc_compress
    for T in (4, 2, 8, 16):
        n = 16 << 20
        data = DATASETS["linspace"](64 << 20)[(4 << 20):(4 << 20) + n]      # the range of the 64 MiB set whose planes change pattern on a row boundary
        r, chunk = orc_compress(oracle, data, T, 5, 1, "lz4")
        assert r > 0
        got_r, got = pkg.decompress(chunk, n)
        assert got_r == n and np.array_equal(got, data), (T, int((got != data).sum()))
