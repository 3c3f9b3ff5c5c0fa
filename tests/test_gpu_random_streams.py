"""GPU: randomly generated, structurally biased LZ4 and BloscLZ streams as the planes of split shuffled blocks, decoded as one batch and compared with the
oracle's reader byte for byte.  Round 5: the silent decode error on the reference's linspace chunks (tests/test_gpu_spans.py) came from a COMBINATION
- a periodic span ending on a row boundary, then a short-period match - that no hand-written case had.  The generator here draws sequences from the
shapes the decoder treats differently (dec_ring.h): literal runs of 0 / a few / hundreds / thousands of bytes; distances 1 .. 65 535 with weight on powers
of two and their neighbours, on "the whole plane so far" and on "just written"; match lengths below 19, with one extension byte, of several rows, of
16 KiB and more (span candidates), and lengths chosen so that the match ENDS or STARTS on a 1 KiB row boundary."""
import ctypes as C

import numpy as np
import pytest

from helpers import ptr, wrap_planes_as_chunk
from test_gpu_decompress import _blz_lits, _blz_match, _lz4_seq, _lz4_tail

pytestmark = pytest.mark.gpu

OFFS = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 512, 1000, 1023, 1024, 1025, 2047, 2048, 2049,
        4096, 8190, 8191, 8192, 8193, 8200, 16384, 32768, 40000, 65535]


def rand_stream(rng, neb, fmt):
    """(stream bytes) that decodes to exactly neb bytes; fmt 1 = LZ4, 0 = BloscLZ"""
    s = bytearray()
    p = 0
    minml = 4 if fmt == 1 else 3
    first = True
    while True:
        room = neb - p - 16                      # bytes that may still be produced in front of the closing literals
        if room < 40:
            break
        u = rng.random()
        ll = 0 if u < 0.5 else int(rng.integers(1, 15)) if u < 0.85 else int(rng.integers(15, 41)) if u < 0.95 else int(rng.integers(100, 3000))
        if first:
            ll = max(ll, int(rng.integers(1, 9)))  # (BloscLZ must start with a literal run; an LZ4 match needs something to copy from)
        ll = min(ll, room - 8)
        q = p + ll                                # where the match starts
        u = rng.random()
        if u < 0.6:
            off = int(rng.choice(OFFS))
        elif u < 0.8:
            off = int(rng.integers(1, q + 1))
        elif u < 0.9:
            off = q                               # from the plane's first byte
        else:
            off = int(rng.integers(1, min(q, 40) + 1))
        off = max(1, min(off, q, 65535))
        u = rng.random()
        if u < 0.5:
            ml = int(rng.integers(minml, 19))
        elif u < 0.75:
            ml = int(rng.integers(19, 274))
        elif u < 0.85:
            ml = int(rng.integers(274, 5000))
        elif u < 0.90:
            ml = int(rng.integers(16000, 70000))
        elif u < 0.96:
            ml = (-q) % 1024 + 1024 * int(rng.integers(0, 40))      # ends exactly on a row boundary
        else:
            ll_fix = (-p) % 1024                                     # starts exactly on a row boundary (the literals bring it there)
            if ll_fix <= room - 8 and (ll_fix > 0 or not first):
                ll = ll_fix; q = p + ll; off = max(1, min(off, q))
            ml = int(rng.integers(minml, 40000))
        ml = max(minml, min(ml, neb - 16 - q))
        if ml < minml or q == 0:
            break
        lit = rng.integers(0, 256, ll, dtype=np.uint8).tobytes()
        if fmt == 1:
            s += _lz4_seq(lit, off, ml)
        else:
            s += _blz_lits(lit) + _blz_match(off, ml)
        p = q + ml
        first = False
    tail = rng.integers(0, 256, neb - p, dtype=np.uint8).tobytes()
    s += _lz4_tail(tail) if fmt == 1 else _blz_lits(tail)
    return bytes(s)


def edge_stream(rng, neb, fmt):
    """Long matches with power-of-two distances <= 1 KiB (dec_ring.h: dr_match's row-register form: the rest of the first row, whole rows and the piece
    behind the last row boundary out of one register set, up to 15 bytes written beyond the match's end), at every alignment of start and end, each
    followed by short sequences whose sources sit at the far edge of the 8 KiB window - the ring slots those extra bytes land in."""
    s = bytearray(); p = 0
    minml = 4 if fmt == 1 else 3

    def emit(lit, off, ml):
        nonlocal s, p
        if fmt == 1: s += _lz4_seq(lit, off, ml)
        else: s += _blz_lits(lit) + _blz_match(off, ml)
        p += len(lit) + ml
    emit(rng.integers(0, 256, int(rng.integers(1100, 1300)), dtype=np.uint8).tobytes(), int(rng.integers(1, 900)), int(rng.integers(minml, 60)))
    while neb - p > 12000:
        off = 1 << int(rng.integers(0, 11))
        ll = int(rng.integers(0, 40)) if rng.random() < 0.7 else int(rng.integers(0, 1100))
        ml = 2048 + int(rng.integers(0, 4200)) if rng.random() < 0.8 else 2048 + 1024 * int(rng.integers(0, 4)) + (-(p + ll)) % 1024
        emit(rng.integers(0, 256, ll, dtype=np.uint8).tobytes(), min(off, p + ll), ml)
        for _ in range(int(rng.integers(1, 5))):                      # sources around the window's far edge, relative to the end of the long match
            ll = int(rng.integers(0, 3))
            off = int(rng.integers(8192 - 40, 8192 + 24)) - int(rng.integers(0, 30))
            emit(rng.integers(0, 256, ll, dtype=np.uint8).tobytes(), max(1, min(off, p + ll)), int(rng.integers(minml, 40)))
    tail = rng.integers(0, 256, neb - p, dtype=np.uint8).tobytes()
    s += _lz4_tail(tail) if fmt == 1 else _blz_lits(tail)
    return bytes(s)


@pytest.mark.parametrize("fmt", [1, 0])
def test_long_power_of_two_matches_and_the_window_edge(pkg, oracle, fmt):
    rng = np.random.default_rng(4242 + fmt)
    bad = []
    for k in range(120):
        T = int(rng.choice([8, 4, 2])); neb = int(rng.choice([128 << 10, 64 << 10, 33 << 10]))
        chunk = wrap_planes_as_chunk([edge_stream(rng, neb, fmt) for _ in range(T)], neb, fmt)
        n = T * neb
        want = np.zeros(n, np.uint8)
        assert oracle.orc_decompress(ptr(chunk), ptr(want), n) == n, k
        r, out = pkg.decompress(chunk, n)
        if r != n or not np.array_equal(out, want):
            bad.append((k, T, neb, r, int((out != want).sum()) if r == n else -1))
    assert not bad, bad[:8]


@pytest.mark.parametrize("fmt,seed", [(1, 1), (1, 2), (0, 3), (0, 4)])
def test_random_streams_as_planes_of_split_blocks(pkg, oracle, fmt, seed):
    rng = np.random.default_rng(1000 + seed)
    chunks, wants = [], []
    for k in range(160):
        T = int(rng.choice([8, 4, 2, 16]))
        neb = int(rng.choice([128 << 10, 128 << 10, 64 << 10, 40 << 10, 17 << 10]))
        streams = [rand_stream(rng, neb, fmt) for _ in range(T)]
        chunk = wrap_planes_as_chunk(streams, neb, fmt)
        n = T * neb
        want = np.zeros(n, np.uint8)
        assert oracle.orc_decompress(ptr(chunk), ptr(want), n) == n, (k, "the generator wrote a stream the reference's reader rejects")
        chunks.append(chunk); wants.append(want)
    n = len(chunks)
    outs = [np.full(w.size, 0xEE, np.uint8) for w in wants]
    L = pkg.load()
    src = (C.c_void_p * n)(*[c.ctypes.data for c in chunks]); dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    ssz = (C.c_size_t * n)(*[c.size for c in chunks]); dsz = (C.c_size_t * n)(*[o.size for o in outs]); res = (C.c_int * n)()
    for rep in range(2):                              # (the second call runs in the cost-feedback order)
        assert L.blosc_gpu_decompress_batch_host(n, src, ssz, dst, dsz, res) == 0
        bad = [(k, res[k], int((outs[k] != wants[k]).sum()), int(np.argmax(outs[k] != wants[k])), int(chunks[k][3])) for k in range(n) if res[k] != wants[k].size or not np.array_equal(outs[k], wants[k])]
        assert not bad, (rep, bad[:8])
