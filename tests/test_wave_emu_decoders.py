"""CPU: the LZ4 and BloscLZ stream decoders of c-blosc_amd/csrc/k_decode.hip (lz4_decode_wave with its batched step and the
LDS-assembled step, blosclz_decode_wave) - the SAME source the GPU runs - executed lane by lane by the wavefront emulator of
tests/tools/wave_emu.  The decoders rely on the wavefront's lock step where one lane loads what another lane stored an
instruction earlier; those places are marked BAMD_MEM_SYNC / BAMD_LDS_SYNC in the source (nothing on the device, a rendezvous
here).  Yardstick: the oracle's decoders (pinned to the reference) - same bytes, same verdict on damaged streams, and never a
byte written outside the room the stream was given."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import DATASETS, ptr
from test_gpu_decompress import _blz_lits, _blz_match, _lz4_seq, _lz4_tail
from test_wave_emu_encoders import emu  # noqa: F401  (fixture: builds tests/tools/liblz_wave_cpu.so)

LZ4, BLOSCLZ = 0, 1
FULL = __import__("os").environ.get("BLOSC_EMU_FULL") == "1"      # the default run is sized for a CPU suite of a few minutes


SOAK = 7919 * int(os.environ.get("BLOSC_EMU_SEED", "0"))      # soak runs (BLOSC_EMU_SEED=1, 2, ...): every random draw of this file moves

def _decode(emu, kind, stream, cap):
    emu.emu_lz_decode.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    s = np.concatenate([np.ascontiguousarray(stream, dtype=np.uint8), np.zeros(256, np.uint8)])     # the engine pads its buffers as well
    out = np.full(cap + 512, 0xEE, np.uint8)
    r = emu.emu_lz_decode(kind, ptr(s), int(np.asarray(stream).size), ptr(out), cap)
    assert np.all(out[cap:] == 0xEE), "wrote outside the room it was given"
    return r, out[:cap]


def _oracle_decode(oracle, kind, stream, cap):
    oracle.orc_blosclz_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    s = np.ascontiguousarray(stream, dtype=np.uint8)
    out = np.full(cap + 8, 0xEE, np.uint8)            # the same fill as _decode: an LZ4 match with offset 0 "copies" what the buffer held (lz4.c:2356)
    f = oracle.orc_blosclz_decompress if kind == BLOSCLZ else oracle.orc_lz4_decompress
    return f(ptr(s), s.size, ptr(out), cap), out[:cap]


def _same_as_oracle(emu, oracle, kind, stream, cap):
    ro, want = _oracle_decode(oracle, kind, stream, cap)
    r, got = _decode(emu, kind, stream, cap)
    ok_o, ok_d = ro == cap and cap > 0, r == cap and cap > 0
    if cap == 0:
        return
    assert ok_o == ok_d, (kind, np.asarray(stream).size, cap, ro, r)
    if ok_o:
        assert np.array_equal(got, want), (kind, np.asarray(stream).size, cap)


def _compress(oracle, kind, data, clevel=5):
    data = np.ascontiguousarray(data)
    dst = np.zeros(data.size + 1024, np.uint8)
    if kind == LZ4:
        r = oracle.orc_lz4_compress(ptr(data), data.size, ptr(dst), data.size - 1, 1)
    else:
        r = oracle.orc_blosclz_compress(clevel, ptr(data), data.size, ptr(dst), data.size - 1, 1)
    return dst[:r].copy() if r > 0 else None


def _inputs(oracle):
    rng = np.random.default_rng(21 + SOAK)
    out = []
    for dname, T in (("bench19", 8), ("linspace", 8), ("randwalk", 8), ("smallints", 4), ("arange", 4)):
        d = DATASETS[dname](T * 16384)
        sh = np.ascontiguousarray(d.reshape(-1, T).T)
        for j in (0, 1, T - 1):
            out.append(sh[j].copy())
        bs = np.zeros(d.size, np.uint8)                               # bit-shuffled: streams full of near matches (the LDS step's case)
        oracle.orc_bitshuffle(T, d.size, ptr(d), ptr(bs))
        out.append(bs[:32768].copy()); out.append(bs[d.size // 2:d.size // 2 + 20000].copy())
    out.append(np.zeros(100000, np.uint8))
    out.append(np.tile(rng.integers(0, 256, 1024, dtype=np.uint8), 40))             # period 1024: the row-replicating copy
    out.append(np.tile(rng.integers(0, 256, 48, dtype=np.uint8), 900))
    out.append(rng.integers(0, 3, 30000, dtype=np.uint8))
    for n in (13, 16, 17, 40, 64, 65, 100, 257, 1000):
        out.append((np.arange(n) % 3).astype(np.uint8))
    return out


@pytest.mark.parametrize("kind", [LZ4, BLOSCLZ], ids=["lz4", "blosclz"])
def test_reference_written_streams(emu, oracle, kind):
    n = 0
    for data in _inputs(oracle):
        s = _compress(oracle, kind, data)
        if s is None:
            continue
        r, got = _decode(emu, kind, s, data.size)
        assert r == data.size and np.array_equal(got, data), (kind, data.size, r)
        n += 1
    assert n > 25


def test_handbuilt_lz4_streams(emu, oracle):
    """The adversarial streams of tests/test_gpu_decompress.py (a thinner grid): overlapping matches at small offsets and at the
    copy-path boundaries, matches that read their own sequence's literals, long literal runs, long runs."""
    rng = np.random.default_rng(11 + SOAK)
    streams = []
    for off in list(range(1, 20)) + [31, 32, 33, 63, 64, 65, 127, 128, 255, 256, 1023, 1024, 1025, 2048, 4099]:
        for mlen in [4, 7, 16, 19, 63, 64, 65, 255, 300, 1024, 1025, 2049, 5000, 70000]:
            s = bytearray()
            pre = rng.integers(0, 256, max(off, 16) + 3, dtype=np.uint8).tobytes()
            s += _lz4_seq(pre, off, mlen)
            s += _lz4_seq(rng.integers(0, 256, 5, dtype=np.uint8).tobytes(), 3, 9)
            s += _lz4_seq(b"", 1, 40)
            s += _lz4_tail(rng.integers(0, 256, 12, dtype=np.uint8).tobytes())
            streams.append(bytes(s))
    for ll in [0, 1, 14, 15, 16, 63, 64, 65, 255, 270, 300, 511, 512, 513, 1024, 5000, 70000]:
        s = bytearray()
        s += _lz4_seq(rng.integers(0, 256, ll + 4, dtype=np.uint8).tobytes(), 4, 20)
        s += _lz4_seq(rng.integers(0, 256, ll, dtype=np.uint8).tobytes(), 7, 4)
        s += _lz4_tail(rng.integers(0, 256, ll + 13, dtype=np.uint8).tobytes())
        streams.append(bytes(s))
    for s in streams:
        s = np.frombuffer(s, np.uint8)
        tmp = np.zeros(1 << 20, np.uint8)
        n = oracle.orc_lz4_decompress(ptr(s), s.size, ptr(tmp), 1 << 20)
        assert n > 0
        for cap in (n, n - 1, n + 1):
            _same_as_oracle(emu, oracle, LZ4, s, cap)


def test_dense_near_match_chains(emu, oracle):
    """Many short sequences per 64 stream bytes whose matches reach into the output of the sequences right before them: the
    LDS-assembled step (lz4_step_lds)."""
    rng = np.random.default_rng(77 + SOAK)

    def chain(nseq, maxoff, maxml, maxll, first_lit):
        s = bytearray(); produced = 0
        lit = rng.integers(0, 256, first_lit, dtype=np.uint8).tobytes()
        for k in range(nseq):
            off = int(rng.integers(1, min(maxoff, produced + len(lit)) + 1))
            ml = int(rng.integers(4, maxml + 1))
            s += _lz4_seq(lit, off, ml)
            produced += len(lit) + ml
            lit = rng.integers(0, 256, int(rng.integers(0, maxll + 1)), dtype=np.uint8).tobytes()
        s += _lz4_tail(rng.integers(0, 256, 12 + int(rng.integers(0, 5)), dtype=np.uint8).tobytes())
        return bytes(s)

    k = 0
    for first_lit in (1, 5, 64, 1030):
        for maxoff in (1, 3, 16, 40, 700, 1024, 1500):
            for maxml, maxll in ((4, 0), (18, 3), (64, 0), (273, 0)):
                s = np.frombuffer(chain(int(rng.integers(20, 200)), maxoff, maxml, maxll, first_lit), np.uint8)
                tmp = np.zeros(1 << 20, np.uint8)
                n = oracle.orc_lz4_decompress(ptr(s), s.size, ptr(tmp), 1 << 20)
                assert n > 0
                _same_as_oracle(emu, oracle, LZ4, s, n)
                k += 1
    assert k == 112


def test_handbuilt_blosclz_streams(emu, oracle):
    rng = np.random.default_rng(12 + SOAK)
    oracle.orc_blosclz_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    streams = []
    for dist in list(range(1, 9)) + [32, 63, 64, 256, 1024, 8191, 8192, 8193, 20000]:          # (the 73 KiB distances stay with the GPU suite: minutes of 32-byte literal runs here)
        for mlen in ([3, 4, 9, 10, 64, 65, 263, 264, 519, 1025, 5000] if dist < 1024 else [3, 264, 5000]):
            s = bytearray()
            s += _blz_lits(rng.integers(0, 256, max(dist, 4) + int(rng.integers(0, 5)), dtype=np.uint8).tobytes())
            s += _blz_match(dist, mlen)
            s += _blz_lits(rng.integers(0, 256, int(rng.integers(1, 33)), dtype=np.uint8).tobytes())
            s += _blz_match(3, 9)
            s += _blz_match(1, 40)
            s += _blz_lits(rng.integers(0, 256, 7, dtype=np.uint8).tobytes())
            streams.append(bytes(s))
    for trial in range(25):
        s = bytearray(_blz_lits(rng.integers(0, 256, 40, dtype=np.uint8).tobytes()))
        produced = 40
        for _ in range(int(rng.integers(20, 300))):
            if rng.random() < 0.35:
                k = int(rng.integers(1, 6)); s += _blz_lits(rng.integers(0, 256, k, dtype=np.uint8).tobytes()); produced += k
            else:
                d = int(rng.integers(1, min(produced, 300) + 1)); m = int(rng.choice([3, 4, 5, 6, 7, 8, 9, 12, 20, 70]))
                s += _blz_match(d, m); produced += m
        s += _blz_lits(rng.integers(0, 256, 9, dtype=np.uint8).tobytes())
        streams.append(bytes(s))
    for s in streams:
        s = np.frombuffer(s, np.uint8)
        tmp = np.zeros(1 << 20, np.uint8)
        n = oracle.orc_blosclz_decompress(ptr(s), s.size, ptr(tmp), 1 << 20)
        assert n > 0
        for cap in (n, n - 1):
            _same_as_oracle(emu, oracle, BLOSCLZ, s, cap)


@pytest.mark.parametrize("kind", [LZ4, BLOSCLZ], ids=["lz4", "blosclz"])
def test_damaged_streams_get_the_oracles_verdict(emu, oracle, kind):
    """Bit flips, truncations and junk: accepted or rejected exactly like the oracle's (= the reference's) decoder, same bytes when
    accepted, nothing written outside the output room (the fuzz rows of SURVEY 8f-1, here without a GPU)."""
    rng = np.random.default_rng(5 + 7919 * int(os.environ.get("BLOSC_EMU_SEED", "0")))      # soak runs: other damage per seed
    tried = 0
    for data in _inputs(oracle)[:14]:
        s = _compress(oracle, kind, data[:6000])
        if s is None or s.size < 20:
            continue
        n = min(data.size, 6000)
        for trial in range(12):
            t = s.copy()
            mode = trial % 4
            if mode == 0:
                t[int(rng.integers(0, t.size))] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                t = t[:int(rng.integers(1, t.size))]
            elif mode == 2:
                pos = int(rng.integers(0, t.size)); t[pos:pos + 4] = rng.integers(0, 256, min(4, t.size - pos), dtype=np.uint8)
            else:
                t = np.concatenate([t, rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8)])
            _same_as_oracle(emu, oracle, kind, t, n)
            tried += 1
    assert tried > 100


# ---- the entropy-coded formats: zlib_decode_wave (k_zlib.hip + inflate_serial.h) and the one-wave Zstd frame decoder (k_zstd.hip) ----
ZSTD, ZLIB = 3, 4


def _entropy_decode(emu, kind, stream, cap):
    emu.emu_entropy_decode.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    s = np.concatenate([np.ascontiguousarray(stream, dtype=np.uint8), np.zeros(256, np.uint8)])
    out = np.full(cap + 512, 0xEE, np.uint8)
    r = emu.emu_entropy_decode(kind, ptr(s), int(np.asarray(stream).size), ptr(out), cap)
    assert np.all(out[cap:] == 0xEE), "wrote outside the room it was given"
    return r, out[:cap]


def test_zlib_streams_written_by_zlib(emu, oracle):
    """Streams of zlib itself (python's): every level, fixed / Huffman-only / RLE strategies, small windows, several blocks,
    stored blocks - and damaged copies of them, which must get the oracle's verdict."""
    from test_oracle_zlib import _un, _zo, stock_streams
    zo = _zo(oracle)
    rng = np.random.default_rng(8 + SOAK)
    n_ok = n_bad = 0
    for s, data in stock_streams(sizes=(1, 17, 255, 1500) if FULL else (17, 1500)):
        r, got = _entropy_decode(emu, ZLIB, s, data.size)
        assert r == data.size and np.array_equal(got, data), (data.size, s.size, r)
        n_ok += 1
        if s.size > 12 and n_ok % 3 == 0:
            for mode in range(3):
                t = s.copy()
                if mode == 0: t[int(rng.integers(2, t.size))] ^= 1 << int(rng.integers(0, 8))
                elif mode == 1: t = t[:int(rng.integers(3, t.size))]
                else: t[-1] ^= 0x10                                           # the Adler-32
                ro, want = _un(zo, t, data.size)
                r, got = _entropy_decode(emu, ZLIB, t, data.size)
                assert (ro == data.size) == (r == data.size), (mode, ro, r)
                if ro == data.size:
                    assert np.array_equal(got, want[:data.size])
                n_bad += 1
    assert n_ok > (100 if FULL else 50) and n_bad > (60 if FULL else 30)


def test_zstd_frames(emu, oracle, ref):
    """Frames of the reference's ZSTD_compress (levels 1 / 3 / 5 / 19: raw, RLE and Huffman literals, predefined / RLE / FSE / repeat
    sequence tables) where oracle/_ref ships, and frames written by this repo's own encoder (all its modes) run on the same emulator."""
    oracle.orc_zstd_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(14 + SOAK)
    inputs = [x for x in _inputs(oracle) if x.size <= 40000][:9 if FULL else 3]
    inputs.append(np.concatenate([DATASETS["bench19"](131072 * 8).reshape(-1, 8).T[1], rng.integers(0, 256, 3000, dtype=np.uint8)]))   # several blocks
    n = 0
    for data in inputs:
        streams = []
        if ref is not None:
            ref.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
            ref.ZSTD_compress.restype = C.c_size_t
            for level in (1, 3, 5, 19):
                dst = np.zeros(data.size + 1024, np.uint8)
                r = ref.ZSTD_compress(ptr(dst), dst.size, ptr(data), data.size, level)
                streams.append(dst[:r].copy())
        from test_wave_emu_encoders import _encode
        for kind in (3, 5, 6, 8, 9):                                         # predefined tables, per-block tables, + search, + Huffman literals
            r, s = _encode(emu, kind, data, clevel=3)
            if r:
                streams.append(s)
        for s in streams:
            r, got = _entropy_decode(emu, ZSTD, s, data.size)
            assert r == data.size and np.array_equal(got, data), (data.size, s.size, r)
            n += 1
            if n % 4 == 0 and s.size > 16:                                     # damaged: the oracle's verdict
                t = s.copy(); t[int(rng.integers(5, t.size))] ^= 1 << int(rng.integers(0, 8))
                want = np.zeros(data.size + 8, np.uint8)
                ro = oracle.orc_zstd_decompress(ptr(t), t.size, ptr(want), data.size)
                r, got = _entropy_decode(emu, ZSTD, t, data.size)
                assert (ro == data.size) == (r == data.size), (ro, r)
                if ro == data.size:
                    assert np.array_equal(got, want[:data.size])
    assert n > (30 if FULL else 10)


# ---- whole split blocks through decode_one_stream: per-stream decode, periodic spans, raw-in-place planes, the fused unshuffle ----
def _block_streams(chunk, T):
    """the T compressed splits of block 0 of a one-block chunk (blosc/blosc.c:635-719 layout: bstarts, then int32 csize + bytes per split)"""
    c = np.asarray(chunk, np.uint8)
    pos = int(c[16:20].view("<i4")[0])
    out = []
    for _ in range(T):
        cs = int(c[pos:pos + 4].view("<i4")[0]); pos += 4
        out.append(c[pos:pos + cs].copy()); pos += cs
    return out


@pytest.mark.parametrize("codec,fmt", [("lz4", 1), ("blosclz", 0)])
@pytest.mark.parametrize("T", [2, 4, 8, 16])
def test_fused_block_decode(emu, oracle, codec, fmt, T):
    """One block = typesize split streams, decoded stream by stream (in shuffled order: any wave may be the one that completes the block)
    by decode_one_stream itself; constant and short-period planes leave only their edges in the scratch (periodic spans, pattern
    table / register row), incompressible planes are read where they lie in the chunk, and the last stream's wave unshuffles."""
    from helpers import header, orc_compress
    emu.emu_decode_block.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_uint, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    rng = np.random.default_rng(30 + T + SOAK)
    ne = 65536                                                     # bytes per plane: long enough for spans (>= 16 KiB matches)
    bsize = ne * T
    seen = {"span": 0, "small": 0, "raw": 0, "plain": 0, "self": 0}
    for trial in range(8 if (FULL or T == 2) else (2 if T == 16 else 4)):      # (typesize 2 needs all eight to see every kind of plane)
        planes = []
        for j in range(T):
            kind = (trial + j) % 8
            if kind == 6:                                                                                    # period above the pattern table: SPAN_SELF, odd start
                per = int(rng.choice([4096, 8192, 16384])); lead = int(rng.integers(0, 9))
                a = np.concatenate([rng.integers(0, 256, lead, dtype=np.uint8), np.resize(rng.integers(0, 256, per, dtype=np.uint8), ne)])[:ne].copy()
                if j % 2 == 0: a[ne - 700:] = a[per + 100:per + 800]                                       # ... and a match reaching back into the span
                planes.append(a)
            elif kind == 7:                                                                                  # runs of period 128 with breaks (the one-load row fill of wave_match_copy)
                a = np.resize(rng.integers(0, 8, 128, dtype=np.uint8), ne).copy()
                for k in range(3000, ne - 200, 4100): a[k:k + 5] = rng.integers(8, 16, 5, dtype=np.uint8); a[k + 5:] = np.resize(np.roll(a[k - 123:k + 5], int(rng.integers(0, 128))), ne - k - 5)
                planes.append(a)
            elif kind == 0: planes.append(np.zeros(ne, np.uint8))                                              # constant: period 1
            elif kind == 1: planes.append(np.resize(rng.integers(0, 256, int(rng.choice([2, 4, 64, 256])), dtype=np.uint8), ne))
            elif kind == 2: planes.append(np.resize(rng.integers(0, 256, int(rng.choice([512, 1024, 2048])), dtype=np.uint8), ne))   # pattern-table span
            elif kind == 3: planes.append(rng.integers(0, 256, ne, dtype=np.uint8))                          # incompressible: stored raw
            elif kind == 4: planes.append(rng.integers(0, 4, ne, dtype=np.uint8))                            # ordinary
            else:                                                                                            # a span that a later match reaches back into
                a = np.resize(rng.integers(0, 256, 128, dtype=np.uint8), ne).copy(); a[ne - 3000:] = a[5000:8000]; a[ne - 100:] = rng.integers(0, 256, 100, dtype=np.uint8)
                planes.append(a)
        data = np.ascontiguousarray(np.stack(planes, 1)).reshape(-1)
        r, chunk = orc_compress(oracle, data, T, 5, 1, codec, blocksize=bsize)
        assert r > 0 and header(chunk)["blocksize"] == bsize and header(chunk)["nbytes"] == bsize
        streams = _block_streams(chunk, T)
        padded = [np.concatenate([s, np.zeros(256, np.uint8)]) for s in streams]
        ptrs = (C.c_void_p * T)(*[p.ctypes.data for p in padded])
        cs = (C.c_int * T)(*[int(s.size) for s in streams])
        order = (C.c_int * T)(*[int(x) for x in rng.permutation(T)])
        dst = np.full(bsize + 256, 0xEE, np.uint8)
        spans = (C.c_uint * 32)()
        st = emu.emu_decode_block(T, fmt, ptrs, cs, bsize, ptr(dst), order, spans)
        assert st == 0
        assert np.array_equal(dst[:bsize], data), (trial, int(np.argmax(dst[:bsize] != data)))
        assert np.all(dst[bsize:] == 0xEE)
        for j in range(T):
            w, hi = spans[2 * j], spans[2 * j + 1]
            if w & 2: seen["raw"] += 1
            elif hi and (w & 4): seen["self"] += 1
            elif hi and (w & 1): seen["small"] += 1
            elif hi: seen["span"] += 1
            else: seen["plain"] += 1
    assert seen["raw"] and seen["small"] and seen["plain"] and (seen["self"] or codec != "lz4"), seen      # (BloscLZ's encoder cuts those planes into shorter matches)
    print(codec, T, seen)


def test_ring_steps_take_reference_streams(emu, oracle, ref):
    """dec_ring.h: the batched steps of the LDS ring decoder must actually run (and be right) on what the benchmark decodes: byte planes
    of bench19 as the reference's own LZ4 writes them (offsets of a few KiB, 13 sequences per 64 stream bytes), a ragged tail, a
    stream that changes between plain steps and long matches, and output rooms too small for the stream; planes longer than the ring
    must take matches out of the rows already written to global memory."""
    emu.emu_ring_steps.restype = C.c_ulonglong
    emu.emu_ring_far.restype = C.c_ulonglong
    far0 = emu.emu_ring_far()
    n = 32768
    planes = DATASETS["bench19"](8 * n).reshape(-1, 8).T.copy()
    before = emu.emu_ring_steps()
    seqs = 0
    for j in (1, 2, 5):
        for cut in (n, n - 1237):
            data = planes[j][:cut].copy()
            stream = _compress(oracle, LZ4, data)
            if stream is None:
                continue
            r, got = _decode(emu, LZ4, stream, data.size)
            assert r == data.size and np.array_equal(got, data), (j, cut)
            for cap in (data.size - 1, data.size // 2, 40, 15):          # too small a room: the oracle's verdict, no byte outside
                _same_as_oracle(emu, oracle, LZ4, stream, cap)
            seqs += 1
    steps = emu.emu_ring_steps() - before
    assert seqs >= 4 and steps >= 50 * seqs, (seqs, steps)
    assert emu.emu_ring_far() - far0 >= 20, emu.emu_ring_far() - far0        # bench19 planes reach back up to 32 KiB: beyond the 8 KiB ring
    # damaged copies of one stream: verdict and bytes of the oracle
    rng = np.random.default_rng(5 + SOAK)
    stream = _compress(oracle, LZ4, planes[1])
    for trial in range(60 if FULL else 25):
        s = stream.copy()
        for pos in rng.integers(0, s.size, 1 + trial % 3):
            s[pos] ^= 1 << int(rng.integers(0, 8))
        _same_as_oracle(emu, oracle, LZ4, s, n)


def test_long_periodic_matches_take_the_row_path(emu, oracle):
    """dec_ring.h: dr_match's path for power-of-two periods (one copy of the period, 16-byte pieces read out of two periods, identical rows
    stored from registers): every period 1 ... 1024, matches of 2 ... 40 rows starting at odd plane positions, several in one stream (the
    ring wraps, rows are flushed in between), and the neighbours of the powers of two (which must NOT take it) - the oracle's bytes."""
    rng = np.random.default_rng(77 + SOAK)
    offs = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 3, 48, 96, 127, 129, 1000, 1023, 1025, 2048]
    for trial, off in enumerate(offs * (2 if FULL else 1)):
        stream = bytearray()
        produced = 0
        first = max(off, 5) + int(rng.integers(0, 700))
        stream += _lz4_seq(rng.integers(0, 256, first, dtype=np.uint8).tobytes(), off, int(rng.integers(2048, 6000)))
        nseq = 3 if off <= 1024 else 2
        plan = [(first, None)]
        # parse back what we wrote to know the output size (simplest: let the oracle say)
        for k in range(nseq):
            lit = rng.integers(0, 256, int(rng.integers(0, 90)), dtype=np.uint8).tobytes()
            o2 = off if k % 2 == 0 else int(rng.choice([1, 4, 128, 256, 1024]))
            stream += _lz4_seq(lit, o2, int(rng.integers(2048, 42000 if k == 1 else 9000)))
        stream += _lz4_tail(rng.integers(0, 256, 12, dtype=np.uint8).tobytes())
        s = np.frombuffer(bytes(stream), np.uint8)
        # the output size: decode once with the oracle into a generous room, shrink to what it produced
        big = 200000
        out = np.zeros(big, np.uint8)
        oracle.orc_lz4_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        # LZ4_decompress_safe wants the exact size: find it by summing the sequences
        def total(b):
            i = 0; n = 0
            while True:
                t = b[i]; i += 1; ll = t >> 4
                if ll == 15:
                    while True:
                        e = b[i]; i += 1; ll += e
                        if e != 255: break
                i += ll; n += ll
                if i >= len(b): return n
                i += 2; ml = t & 15
                if ml == 15:
                    while True:
                        e = b[i]; i += 1; ml += e
                        if e != 255: break
                n += ml + 4
        cap = total(bytes(stream))
        _same_as_oracle(emu, oracle, LZ4, s, cap)
        r, got = _decode(emu, LZ4, s, cap)
        assert r == cap, (off, r, cap)
