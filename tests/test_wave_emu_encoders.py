"""CPU: the wave-parallel encoders of c-blosc_amd/csrc/k_encode.hip - lz_encode_wave (LZ4, BloscLZ), lz4hc_encode_wave
(the LZ4HC-grade search), zstd_encode_wave (predefined and per-block sequence tables) and zlib_encode_wave - the SAME
source the GPU runs, executed lane by lane on the CPU by the wavefront emulator of tests/tools/wave_emu (every
cross-lane instruction is a rendezvous of 64 coroutines; BAMD_LDS_SYNC marks where lanes talk through LDS).  What they
write must be a valid stream: the oracle's LZ4 / BloscLZ decoders (pinned to the reference) and, where oracle/_ref
ships, the reference's own decoders read every one of them back bit-exactly; a cross-lane instruction placed in
divergent control flow aborts the run.  The GPU suite checks the same functions on the device; this file is what keeps
them honest between GPU runs (and what a change to the match finder is tried against first)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import DATASETS, ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
FULL = os.environ.get("BLOSC_EMU_FULL") == "1"          # the default run is sized for a CPU suite of a few minutes; BLOSC_EMU_FULL=1 takes everything
LZ4, BLOSCLZ, LZ4HC, ZSTD, ZLIB, ZSTD_TABLES, ZSTD_SEARCH, ZLIB_SEARCH, ZSTD_HUF, ZSTD_SEARCH_HUF, ZLIB_DYN, ZLIB_DYN_SEARCH = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
# the LZ4 front ends by name (kind 0 = whichever the level picks): the sequential select / emit loop of enc_lz.h, and the parallel parse of enc_lz4p.h probing
# every position / every other position
LZ4_SEQ, LZ4_PAR1, LZ4_PAR2 = 12, 13, 14


SOAK = 7919 * int(os.environ.get("BLOSC_EMU_SEED", "0"))      # soak runs (BLOSC_EMU_SEED=1, 2, ...): every random draw of this file moves

@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("the emulator needs the ROCm clang++ (ext_vector_type / address_space in host code)")
    csrc = os.path.join(ROOT, "c-blosc_amd", "csrc")
    tools = os.path.join(ROOT, "tests", "tools")
    so = os.path.join(tools, "liblz_wave_cpu.so")
    deps = [os.path.join(tools, "lz_wave_cpu.cpp"), os.path.join(tools, "wave_emu", "wave_emu.h")]
    deps += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    if os.environ.get("BLOSC_WAVE_EMU_LIB"):     # a build of one's own (sanitizers: HISTORY.md, round 5)
        so = os.environ["BLOSC_WAVE_EMU_LIB"]
    elif not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([CLANG, "-std=c++17", "-O1", "-shared", "-fPIC", "-w", "-I", os.path.join(tools, "wave_emu"), "-I", csrc,
                               "-x", "c++", deps[0], "-o", so])
    E = C.CDLL(so)
    E.emu_lz_encode.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
    return E


def _encode(emu, kind, data, cap=None, clevel=9):
    data = np.ascontiguousarray(data)
    cap = data.size if cap is None else cap
    dst = np.full(max(cap, 1) + 64, 0xEE, np.uint8)
    r = emu.emu_lz_encode(kind, ptr(data), data.size, ptr(dst), cap, clevel, None)
    assert 0 <= r <= cap, (r, cap)
    assert np.all(dst[max(cap, 1):] == 0xEE), "wrote beyond the capacity it was given"
    return r, dst[:r].copy()


def _decodes(oracle, ref, kind, stream, data):
    n = data.size
    back = np.zeros(n + 8, np.uint8)
    if kind == BLOSCLZ:
        assert oracle.orc_blosclz_decompress(ptr(stream), stream.size, ptr(back), n) == n
    else:
        assert oracle.orc_lz4_decompress(ptr(stream), stream.size, ptr(back), n) == n
    assert np.array_equal(back[:n], data)
    if ref is not None:
        back2 = np.zeros(n + 8, np.uint8)
        if kind == BLOSCLZ:
            assert ref.blosclz_decompress(ptr(stream), stream.size, ptr(back2), n) == n
        else:
            assert ref.LZ4_decompress_safe(ptr(stream), ptr(back2), stream.size, n) == n
        assert np.array_equal(back2[:n], data)


def _plane(dname, n, T=8, j=0):
    d = DATASETS[dname](n * T)
    return np.ascontiguousarray(d.reshape(-1, T).T[j])


@pytest.mark.parametrize("kind", [LZ4, BLOSCLZ, LZ4HC, LZ4_SEQ, LZ4_PAR1, LZ4_PAR2], ids=["lz4", "blosclz", "lz4hc", "lz4-sequential", "lz4-parallel-stride1", "lz4-parallel-stride2"])
def test_streams_decode_with_oracle_and_reference(emu, oracle, ref, kind):
    oracle.orc_blosclz_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(5 + SOAK)
    cases = 0
    inputs = []
    for dname in ["bench19", "linspace", "randwalk", "smallints", "arange"]:
        for j in (0, 1, 3):
            for n in (4096, 16384 + 64):
                inputs.append(_plane(dname, n, 8, j))
        inputs.append(DATASETS[dname](8192))                          # unshuffled
    for n in [0, 1, 4, 12, 13, 14, 15, 16, 17, 19, 20, 21, 31, 32, 63, 64, 65, 66, 76, 77, 127, 128, 129, 130, 191, 255, 256, 257, 300, 1000, 1023, 1024, 1025]:
        inputs.append(np.zeros(n, np.uint8))
        inputs.append((np.arange(n) % 7).astype(np.uint8))
        inputs.append(rng.integers(0, 4, n, dtype=np.uint8))
        inputs.append(rng.integers(0, 256, n, dtype=np.uint8))
    inputs.append(np.zeros(200000, np.uint8))                         # one long run
    inputs.append(np.tile(rng.integers(0, 256, 300, dtype=np.uint8), 100))          # one long far-period match
    inputs.append(np.concatenate([rng.integers(0, 256, 70000, dtype=np.uint8)] * 2))   # a match beyond 64 KiB: must not be used (LZ4) / far (BloscLZ)
    inputs.append(np.concatenate([np.tile(rng.integers(0, 256, 40, dtype=np.uint8), 30), rng.integers(0, 256, 500, dtype=np.uint8)] * 8))
    for data in (inputs if FULL else inputs[::2]):
        for clevel in ((9,) if kind == LZ4HC else ((1, 5, 9) if FULL else (1, 9))):
            r, s = _encode(emu, kind, data, clevel=clevel)
            if r == 0:
                continue                                              # "store raw": the caller copies the plane
            assert r < data.size
            _decodes(oracle, ref, kind, s, data)
            cases += 1
    assert cases > (60 if FULL else 25)


@pytest.mark.parametrize("kind", [LZ4, LZ4HC, LZ4_PAR1, LZ4_PAR2], ids=["lz4", "lz4hc", "lz4-parallel-stride1", "lz4-parallel-stride2"])
def test_capacity_is_respected(emu, oracle, ref, kind):
    """Whatever room the stream is given: either a complete valid stream inside it, or 0 (tests/test_maxout.c's rule one level down)."""
    for data in (_plane("bench19", 8192, 8, 0), _plane("bench19", 8192, 8, 1), _plane("linspace", 8192, 8, 2), _plane("linspace", 8192, 8, 6)):
        full, _ = _encode(emu, kind, data)
        assert full > 0
        for cap in [0, 1, 5, 12, 13, 20, full - 40, full - 9, full - 1, full, full + 1, full + 8, data.size]:
            if cap < 0:
                continue
            r, s = _encode(emu, kind, data, cap=cap)
            if r:
                _decodes(oracle, ref, kind, s, data)
            if cap >= full + 8:
                assert r == full


def _lz_source(rng, n):
    """Bytes made the way an LZ decoder makes them - literal runs over a small alphabet and copies from the history at short and long distances - so that
    the match finder meets matches of every length around its limits (4, 7, RANK_CAP, a step's 64 / 128 positions), at every parity and alignment."""
    out = bytearray()
    alpha = int(rng.choice([2, 4, 16, 256]))
    while len(out) < n:
        if len(out) < 4 or rng.random() < 0.35:
            out += bytes(rng.integers(0, alpha, int(rng.choice([1, 2, 3, 5, 9, 17, 40, 130])), dtype=np.uint8))
        else:
            off = int(min(len(out), rng.choice([1, 2, 3, 4, 7, 8, 16, 31, 64, 100, 255, 1000, 5000, 40000])))
            for _ in range(int(rng.choice([3, 4, 5, 6, 7, 8, 9, 12, 19, 20, 21, 24, 40, 64, 65, 127, 128, 129, 300, 2000]))):
                out.append(out[len(out) - off])
    return np.frombuffer(bytes(out[:n]), np.uint8).copy()


@pytest.mark.parametrize("kind", [LZ4_PAR1, LZ4_PAR2], ids=["stride1", "stride2"])
def test_parallel_parse_on_random_lz_sources(emu, oracle, ref, kind):
    """The parallel parse (enc_lz4p.h) on seeded random sources, sizes and capacities: a valid stream inside the room it was given, or 0.  (Round 6: the
    two-positions-per-lane step started a chain behind the last probing lane - a sequence made of another lane's leftovers, found by exactly this kind of input.)"""
    rng = np.random.default_rng(77 + SOAK)
    cases = 0
    for _ in range(120 if FULL else 40):
        n = int(rng.choice([13, 14, 20, 64, 65, 100, 127, 128, 129, 130, 200, 255, 256, 257, 300, 511, 513, 1000, 4096, 5000, 20000, 70000])) + int(rng.integers(0, 3))
        data = _lz_source(rng, n)
        for clevel in (1, 5, 9):      # (9: shortest match 4 - where "found one byte late" must not turn a 3-byte match into a sequence: caught by the FULL run of round 6)
            cap = n if rng.random() < 0.7 else int(rng.integers(0, n + 1))
            r, s = _encode(emu, kind, data, cap=cap, clevel=clevel)
            if r:
                _decodes(oracle, ref, LZ4, s, data)
                cases += 1
    assert cases > 25


def test_stride2_keeps_most_of_the_ratio(emu):
    """Probing every other position (clevel <= 5) may cost a few per cent of ratio on the benchmark's planes, not more (bench19 at typesize 8: 47.8 -> 46.6 here,
    the reference's LZ4_compress_fast at its acceleration for that level: 36.7)."""
    tot = {LZ4_PAR1: 0, LZ4_PAR2: 0}
    n = 0
    for j in range(8):
        data = _plane("bench19", 131072, 8, j)
        n += data.size
        for kind in tot:
            r, _ = _encode(emu, kind, data, clevel=5)
            tot[kind] += r or data.size
    print(f"bench19 planes of 128 KiB: every position {n / tot[LZ4_PAR1]:.2f}  every other position {n / tot[LZ4_PAR2]:.2f}")
    assert tot[LZ4_PAR2] <= tot[LZ4_PAR1] * 1.06


def test_lz4hc_search_finds_what_the_plain_one_misses(emu, oracle, ref):
    """The point of the search: the ratio.  Against the reference's LZ4_compress_HC (level 9) where oracle/_ref ships, and
    always against the plain LZ4 match finder on the same planes (tests/tools/enc_model2.c is the model behind the numbers)."""
    tot = {"plain": 0, "hc": 0, "ref": 0, "n": 0}
    for dname, T in (("bench19", 8), ("bench19", 4), ("linspace", 8)):
        for j in range(0, T, 2):
            data = _plane(dname, 32768, T, j)
            a, _ = _encode(emu, LZ4, data)
            h, s = _encode(emu, LZ4HC, data)
            a = a or data.size; h = h or data.size
            tot["plain"] += a; tot["hc"] += h; tot["n"] += data.size
            if ref is not None:
                ref.LZ4_compress_HC.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
                out = np.zeros(data.size + 64, np.uint8)
                rr = ref.LZ4_compress_HC(ptr(data), ptr(out), data.size, data.size, 9)
                tot["ref"] += rr if rr > 0 else data.size
    print(f"planes of 32 KiB: plain {tot['n'] / tot['plain']:.2f}  lz4hc search {tot['n'] / tot['hc']:.2f}" + (f"  LZ4_compress_HC(9) {tot['n'] / tot['ref']:.2f}" if tot["ref"] else ""))
    assert tot["hc"] <= tot["plain"] * 0.93
    if tot["ref"]:
        assert tot["hc"] <= tot["ref"] * 1.10


def _zstd_reads(oracle, ref, stream, data):
    n = data.size
    oracle.orc_zstd_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    back = np.zeros(n + 8, np.uint8)
    assert oracle.orc_zstd_decompress(ptr(stream), stream.size, ptr(back), n) == n and np.array_equal(back[:n], data)
    if ref is not None:
        ref.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        ref.ZSTD_decompress.restype = C.c_size_t
        back2 = np.zeros(n + 8, np.uint8)
        assert ref.ZSTD_decompress(ptr(back2), n, ptr(stream), stream.size) == n and np.array_equal(back2[:n], data)


def _entropy_inputs():
    rng = np.random.default_rng(9 + SOAK)
    inputs = []
    for dname in ["bench19", "linspace", "randwalk", "smallints", "arange"]:
        for j in (1, 5):
            inputs.append(_plane(dname, 8192, 8, j))
    inputs.append(np.concatenate([_plane("bench19", 131072, 8, 1)[:131072 - 3000], _plane("linspace", 8192, 8, 1)]))   # two blocks in one frame: the second one starts on a table the first one's scratch has overwritten
    inputs.append(np.concatenate([_plane("bench19", 131072, 8, 1), rng.integers(0, 256, 1069, dtype=np.uint8)]))   # ... and a second block without a single sequence (stored raw)
    for n in [0, 1, 16, 31, 32, 33, 63, 64, 65, 100, 255, 256, 1000, 4097]:
        inputs.append(np.zeros(n, np.uint8))                           # one sequence: every alphabet RLE
        inputs.append((np.arange(n) % 5).astype(np.uint8))
        inputs.append(rng.integers(0, 3, n, dtype=np.uint8))
        inputs.append(rng.integers(0, 256, n, dtype=np.uint8))
    inputs.append(np.concatenate([np.tile(rng.integers(0, 256, int(k), dtype=np.uint8), 40) for k in rng.integers(3, 200, 60)]))   # many match-length / offset codes
    inputs.append(np.concatenate([rng.integers(0, 256, int(k), dtype=np.uint8) if i % 2 else np.zeros(int(k), np.uint8) for i, k in enumerate(rng.integers(1, 90, 400))]))  # many literal-length codes
    return inputs


@pytest.mark.parametrize("kind", [ZSTD, ZSTD_TABLES, ZSTD_SEARCH, ZSTD_HUF, ZSTD_SEARCH_HUF],
                         ids=["predefined", "per-block-tables", "tables+lz4hc-search", "tables+huffman-literals", "tables+search+huffman"])
def test_zstd_frames_decode_with_oracle_and_reference(emu, oracle, ref, kind):
    cases = 0
    inputs = _entropy_inputs()
    for data in (inputs if (kind == ZSTD_TABLES and FULL) else inputs[::2]):
        for clevel in ((1, 9) if kind == ZSTD else (3,)):
            r, s = _encode(emu, kind, data, clevel=clevel)
            if r:
                assert r < data.size
                _zstd_reads(oracle, ref, s, data)
                cases += 1
    assert cases > 8
    # capacity: a complete frame inside what it was given, or 0
    data = _plane("bench19", 16384, 8, 1)
    full, _ = _encode(emu, kind, data, clevel=3)
    assert full > 0
    for cap in [0, 20, 63, 64, 100, full - 30, full - 1, full, full + 1, full + 40, data.size]:
        r, s = _encode(emu, kind, data, cap=max(cap, 0), clevel=3)
        if r:
            _zstd_reads(oracle, ref, s, data)


def test_zstd_tables_made_for_the_block_pay(emu, oracle, ref):
    """Sizes with per-block tables against the predefined ones on the SURVEY 8d planes (16 KiB: what a 64 MiB chunk's blocks
    look like at clevel 3): never larger by more than a header's worth, and on bench19 / linspace at least 20 % smaller."""
    for dname, want in (("bench19", 0.83), ("linspace", 0.83), ("randwalk", 1.001), ("smallints", 1.001)):
        a = b = 0
        for j in range(8):
            data = _plane(dname, 16384, 8, j)
            ra, _ = _encode(emu, ZSTD, data, clevel=3)
            rb, s = _encode(emu, ZSTD_TABLES, data, clevel=3)
            a += ra or data.size; b += rb or data.size
        print(f"{dname}: predefined tables {8 * 16384 / a:.2f}, per-block tables {8 * 16384 / b:.2f}")
        assert b <= a * want, (dname, a, b)


@pytest.mark.parametrize("kind", [ZLIB, ZLIB_SEARCH, ZLIB_DYN, ZLIB_DYN_SEARCH], ids=["plain", "lz4hc-search", "dynamic-codes", "dynamic-codes+search"])
def test_zlib_streams_decode(emu, oracle, ref, kind):
    import zlib
    cases = 0
    for data in (_entropy_inputs() if FULL else _entropy_inputs()[::2]):
        for clevel in ((1, 5, 9) if (kind == ZLIB and FULL) else (5,)):
            r, s = _encode(emu, kind, data, clevel=clevel)
            if r:
                assert zlib.decompress(s.tobytes()) == data.tobytes()
                cases += 1
    assert cases > 12


def test_zstd_huffman_literals(emu, oracle, ref):
    """Literal-heavy inputs: alphabets of 2 .. 256 byte values, flat and extremely skewed (the 11-bit limit and the repair of the code
    space), values above 128 (the weights travel FSE-compressed), runs of 64 .. 130 000 literals (one stream / four streams, the three
    header sizes).  Every frame is read by the oracle and ZSTD_decompress; where the bytes are compressible as literals the frames must
    be smaller than with raw literals."""
    rng = np.random.default_rng(3 + SOAK)
    smaller = tried = 0
    for trial in range(16):
        n = int(rng.choice([255, 256, 300, 1000, 1023, 1024, 5000, 16383, 16384, 20000]))
        k = int(rng.choice([2, 3, 5, 16, 100, 129, 200, 256]))
        pr = np.random.default_rng(trial).dirichlet(np.ones(k) * rng.choice([0.02, 0.5, 5]))
        data = rng.choice(k, n, p=pr).astype(np.uint8)
        if trial % 3 == 0:
            data = (data.astype(np.int32) * int(rng.integers(1, 256 // k + 1))).astype(np.uint8)
        ra, _ = _encode(emu, ZSTD_TABLES, data, clevel=3)
        rb, s = _encode(emu, ZSTD_HUF, data, clevel=3)
        if rb:
            _zstd_reads(oracle, ref, s, data)
            tried += 1
            assert rb <= (ra or data.size)
            smaller += rb < (ra or data.size)
    assert tried > (0 if SOAK else 6) and smaller > (0 if SOAK else 5)          # (how many draws are compressible as literals depends on the draws: the count is pinned for the default seed only)
    for dname, T, want in (("smallints", 4, 0.90), ("randwalk", 8, 1.001)):
        d = DATASETS[dname](131072)
        block = np.ascontiguousarray(d.reshape(-1, T).T).reshape(-1)           # an unsplit shuffled block, as blosc hands it to Zstd
        ra, _ = _encode(emu, ZSTD_TABLES, block, clevel=3)
        rb, s = _encode(emu, ZSTD_HUF, block, clevel=3)
        _zstd_reads(oracle, ref, s, block)
        print(f"{dname}: raw literals {block.size / ra:.2f}, Huffman literals {block.size / rb:.2f}")
        assert rb <= ra * want


def test_zlib_dynamic_codes_pay(emu):
    """Streams with codes of their own against the fixed codes, on unsplit shuffled blocks (what blosc hands to zlib): never larger,
    and clearly smaller where the symbols are unevenly used; python's zlib reads them."""
    import zlib
    for dname, T, want in (("linspace", 8, 0.93), ("smallints", 4, 0.90), ("randwalk", 8, 0.99), ("bench19", 8, 1.0)):
        d = DATASETS[dname](65536)
        block = np.ascontiguousarray(d.reshape(-1, T).T).reshape(-1)
        ra, _ = _encode(emu, ZLIB, block, clevel=5)
        rb, s = _encode(emu, ZLIB_DYN, block, clevel=5)
        assert zlib.decompress(s.tobytes()) == block.tobytes()
        print(f"{dname}: fixed codes {block.size / (ra or block.size):.2f}, dynamic codes {block.size / rb:.2f}")
        assert rb <= (ra or block.size) * want, (dname, ra, rb)
    # exactly 256 KiB: blosc's zlib block size at clevel 5 for typesize 4 (the triples' 18-bit fields must hold it)
    d = DATASETS["smallints"](262144)
    block = np.ascontiguousarray(d.reshape(-1, 4).T).reshape(-1)
    rb, s = _encode(emu, ZLIB_DYN, block, clevel=5)
    assert 0 < rb < block.size * 0.6 and zlib.decompress(s.tobytes()) == block.tobytes()
    # odd symbol statistics: one literal value, no match at all, one distance only, every length symbol
    rng = np.random.default_rng(6 + SOAK)
    odd = [np.zeros(5000, np.uint8), rng.integers(0, 256, 3000, dtype=np.uint8), np.tile(rng.integers(0, 256, 700, dtype=np.uint8), 30),
           np.concatenate([np.tile(rng.integers(0, 256, int(k), dtype=np.uint8), 3) for k in range(3, 300)]),
           rng.choice(3, 20000, p=[0.98, 0.01, 0.01]).astype(np.uint8), np.arange(70000, dtype=np.uint32).view(np.uint8)[:70000]]
    for data in odd:
        for kind in (ZLIB_DYN, ZLIB_DYN_SEARCH):
            r, s = _encode(emu, kind, data, clevel=5)
            if r:
                assert zlib.decompress(s.tobytes()) == data.tobytes()
