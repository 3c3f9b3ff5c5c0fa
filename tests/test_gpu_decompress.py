"""GPU decompress path (plan -> LZ decode -> unshuffle) is bit-exact on chunks written by the
reference algorithm (oracle, pinned to the real reference) and on the reference's golden vectors."""
import glob
import os

import ctypes as C

import numpy as np
import pytest

from helpers import DATASETS, orc_compress, orc_decompress, header, ptr, wrap_stream_as_chunk, wrap_planes_as_chunk

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("fname", sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "compat", "*.cdata"))))
def test_compat_golden_vectors(pkg, fname):
    """compat/*.cdata (blosc 1.3.0 ... 1.18.0 writers): BloscLZ / LZ4 / LZ4HC decode to arange(1e6, int32);
    Zstd (k_zstd.hip) and Zlib (k_zlib.hip) too; Snappy gives -5 (not built in), as stock does (compat/filegen.c:97-103)."""
    chunk = np.fromfile(os.path.join(GOLDEN, "compat", fname), np.uint8)
    r, out = pkg.decompress(chunk, 4000000)
    if "snappy" in fname:
        assert r == -5
    else:
        assert r == 4000000
        assert np.array_equal(out.view("<i4"), np.arange(10**6, dtype="<i4"))


@pytest.mark.parametrize("codec", ["lz4", "blosclz"])
@pytest.mark.parametrize("shuffle", [0, 1, 2])
def test_decode_oracle_chunks(pkg, oracle, codec, shuffle):
    rng_sizes = [0, 1, 15, 127, 128, 129, 1000, 4096, 32768, 65536 + 17, 300001, (1 << 20), (1 << 21) + 5, 641091]
    bad = []
    for T in [1, 2, 3, 4, 7, 8, 16, 17, 32]:
        for n in rng_sizes:
            for dname in ["bench19", "randwalk", "zeros", "smallints", "random"]:
                if n > 400000 and dname in ("random", "zeros") and T not in (4, 8):
                    continue
                data = DATASETS[dname](n)
                for clevel in ([1, 5, 9] if n < 400000 else [5]):
                    r, chunk = orc_compress(oracle, data, T, clevel, shuffle, codec)
                    assert r > 0
                    r2, out = pkg.decompress(chunk, n)
                    if r2 != n or not np.array_equal(out, data):
                        bad.append((T, n, dname, clevel, r2))
    assert not bad, bad[:10]


def test_forced_blocksizes_and_splitmodes(pkg, oracle):
    bad = []
    for T in [4, 8, 16]:
        for bs in [128, 1000, 4096 + T, 65536, 1 << 18]:
            for sm in [1, 2, 3, 4]:
                data = DATASETS["bench19"](500000)
                r, chunk = orc_compress(oracle, data, T, 5, 1, "lz4", blocksize=bs, splitmode=sm)
                r2, out = pkg.decompress(chunk, data.size)
                if r2 != data.size or not np.array_equal(out, data):
                    bad.append((T, bs, sm, r2))
    assert not bad, bad


def test_memcpyed_and_empty(pkg, oracle):
    for n in [0, 1, 100, 127]:
        data = DATASETS["random"](n)
        r, chunk = orc_compress(oracle, data, 4, 5, 1, "lz4")
        assert r == n + 16
        r2, out = pkg.decompress(chunk, n)
        assert r2 == n and np.array_equal(out, data)
    data = DATASETS["random"](1 << 20)   # incompressible -> MEMCPYED fallback chunk
    r, chunk = orc_compress(oracle, data, 8, 5, 1, "lz4")
    assert header(chunk)["flags"] & 2
    r2, out = pkg.decompress(chunk, data.size)
    assert r2 == data.size and np.array_equal(out, data)
    data = DATASETS["bench19"](1 << 20)  # clevel 0
    r, chunk = orc_compress(oracle, data, 8, 0, 1, "lz4")
    r2, out = pkg.decompress(chunk, data.size)
    assert r2 == data.size and np.array_equal(out, data)


def test_error_returns_match_oracle(pkg, oracle, lib):
    """Header / chain validation (blosc.c:1463-1508, :762-770): same negative codes as the oracle."""
    data = DATASETS["bench19"](300000)
    _, good = orc_compress(oracle, data, 4, 5, 1, "lz4")
    cases = []
    c = good.copy(); c[0] = 3; cases.append(("version", c))
    c = good.copy(); c[1] = 2; cases.append(("versionlz", c))
    c = good.copy(); c[2] |= 0x08; cases.append(("reserved flag", c))
    c = good.copy(); c[2] = (c[2] & 0x1f) | (2 << 5); cases.append(("snappy format", c))
    c = good.copy(); c[2] = (c[2] & 0x1f) | (3 << 5); cases.append(("zlib format over LZ4 streams", c))
    c = good.copy(); c[8:12] = 0; cases.append(("blocksize 0", c))
    c = good.copy(); c[12:16] = np.array([20], "<i4").view(np.uint8); cases.append(("cbytes too small", c))
    c = good.copy(); c[16:20] = np.array([10**9], "<i4").view(np.uint8); cases.append(("bstart oob", c))
    c = good.copy(); c[16:20] = np.array([-5], "<i4").view(np.uint8); cases.append(("bstart negative", c))
    hb = header(good); first = int(good[16:20].view("<i4")[0])
    c = good.copy(); c[first:first + 4] = np.array([10**8], "<i4").view(np.uint8); cases.append(("csize oob", c))
    for name, ch in cases:
        want = oracle.orc_decompress(ptr(ch), ptr(np.zeros(data.size, np.uint8)), data.size)
        got, _ = pkg.decompress(ch, data.size)
        assert got == want and got < 0, (name, got, want)
    # destination too small
    got = lib.blosc_decompress_ctx(ptr(good), ptr(np.zeros(data.size, np.uint8)), data.size - 1, 1)
    assert got == -1


def test_negative_nbytes_gets_the_references_verdict(pkg, oracle, lib, ref):
    """A header whose nbytes has bit 31 set (damaged input only): the reference counts nbytes / blocksize <= 0 blocks, runs none and returns 0 with
    nothing written, unless one of its remaining header checks fails first (blosc.c:1463-1511).  Rounds 1 - 5 answered -1 (a stated deviation, closed in
    round 6): the verdicts must be EQUAL, through the stock entry point and through the batched device-resident one."""
    import torch
    assert ref is not None, "oracle/_ref must travel to the GPU box"
    data = DATASETS["bench19"](300000)
    for cname in ("lz4", "blosclz"):
        _, good = orc_compress(oracle, data, 8, 5, 1, cname)
        base = good.copy(); base[7] ^= 0x80
        variants = [("plain", base)]
        t = base.copy(); t[4:8] = np.array([-1], "<i4").view(np.uint8); variants.append(("nbytes -1", t))
        t = base.copy(); t[4:8] = np.array([-(1 << 31)], "<i4").view(np.uint8); variants.append(("nbytes INT_MIN", t))
        t = base.copy(); t[2] |= 0x02; variants.append(("memcpyed flag, sizes disagree", t))
        t = base.copy(); t[2] |= 0x02; t[12:16] = (t[4:8].view("<i4") + 16).view(np.uint8); variants.append(("memcpyed flag, sizes agree", t))
        t = base.copy(); t[2] = (t[2] & 0x1f) | (2 << 5); variants.append(("snappy format id", t))
        t = base.copy(); t[1] = 7; variants.append(("versionlz", t))
        t = base.copy(); t[12:16] = np.array([0], "<i4").view(np.uint8); variants.append(("cbytes 0", t))
        t = base.copy(); t[0] = 3; variants.append(("version", t))
        t = base.copy(); t[2] |= 0x08; variants.append(("reserved flag", t))
        t = base.copy(); t[8:12] = np.array([64], "<i4").view(np.uint8); variants.append(("small blocksize", t))
        want = []
        for name, t in variants:
            dst = np.full(data.size, 0xEE, np.uint8)
            rr = ref.blosc_decompress_ctx(ptr(t), ptr(dst), data.size, 1)
            assert np.all(dst == 0xEE)
            want.append(rr)
            dst = np.full(data.size, 0xEE, np.uint8)
            rg = lib.blosc_decompress_ctx(ptr(t), ptr(dst), data.size, 1)
            assert rg == rr and np.all(dst == 0xEE), (cname, name, rg, rr)
        assert want[0] == 0
        dev = torch.device("cuda:0")
        d_src = [torch.from_numpy(t).to(dev) for _, t in variants]
        d_dst = [torch.full((data.size,), 0xA5, dtype=torch.uint8, device=dev) for _ in variants]
        b = pkg.DeviceBatch([t.data_ptr() for t in d_src], [t.size for _, t in variants], [t.data_ptr() for t in d_dst], [data.size] * len(variants))
        assert b.decompress(with_srcsize=False) == 0          # (with source sizes the batched call also rejects a cbytes beyond the buffer: an extension the reference has no argument for)
        assert list(b.results()) == want, (cname, list(b.results()), want)
        for t in d_dst: assert bool((t == 0xA5).all())


def test_corrupt_payload_same_verdict_and_bytes_as_oracle(pkg, oracle):
    """Random damage to LZ4 / BloscLZ chunks, device-resident, output in canary-padded buffers: the verdict equals
    the oracle's (itself pinned to the reference), accepted chunks carry the oracle's bytes, and nothing is
    written outside [dest, dest + nbytes).  (blosc.h:258-263 promises memory safety; parity asks for more.)"""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    PAD = 4096
    oracle.orc_lz4_offset0_seen.restype = C.c_long
    for dname, T, shuffle in [("bench19", 8, 1), ("bench19", 4, 0), ("linspace", 8, 1), ("smallints", 4, 2)]:
        data = DATASETS[dname](1 << 20)
        for codec in ["lz4", "blosclz"]:
            _, good = orc_compress(oracle, data, T, 5, shuffle, codec)
            first = 16 + 4 * ((data.size + header(good)["blocksize"] - 1) // header(good)["blocksize"])
            chunks, want = [], []
            for trial in range(40):
                c = good.copy()
                k = int(rng.integers(1, 12))
                pos = rng.integers(first, c.size, k)
                c[pos] = rng.integers(0, 256, k, dtype=np.uint8)
                oracle.orc_lz4_offset0_seen(1)
                ro, oo = orc_decompress(oracle, c, data.size)
                want.append((ro, oo.copy(), oracle.orc_lz4_offset0_seen(1)))
                chunks.append(c)
            d_src = [torch.from_numpy(c).to(dev) for c in chunks]
            d_dst = [torch.full((data.size + 2 * PAD,), 0xA5, dtype=torch.uint8, device=dev) for _ in chunks]
            b = pkg.DeviceBatch([t.data_ptr() for t in d_src], [c.size for c in chunks],
                                [t.data_ptr() + PAD for t in d_dst], [data.size] * len(chunks))
            assert b.decompress() == 0
            for i, (rg, (ro, oo, off0)) in enumerate(zip(b.results(), want)):
                out = d_dst[i].cpu().numpy()
                assert (out[:PAD] == 0xA5).all() and (out[PAD + data.size:] == 0xA5).all(), (dname, codec, i, "canary")
                if ro == data.size:
                    assert rg == data.size, (dname, codec, i, rg, ro)
                    if not off0:       # an offset-0 LZ4 match leaves unspecified bytes in the reference too (lz4.c:2356)
                        assert np.array_equal(out[PAD:PAD + data.size], oo), (dname, codec, i)
                else:
                    assert rg < 0, (dname, codec, i, rg, ro)
            # an intact chunk still decodes afterwards
            r, out = pkg.decompress(good, data.size)
            assert r == data.size and np.array_equal(out, data)


def _lz4_seq(lit, off, mlen):
    """one LZ4 sequence (token, ext, literals, offset, ext)"""
    out = bytearray()
    ll = len(lit); mc = mlen - 4
    out.append((min(ll, 15) << 4) | min(mc, 15))
    if ll >= 15:
        v = ll - 15
        while v >= 255: out.append(255); v -= 255
        out.append(v)
    out += bytes(lit)
    out += bytes([off & 255, off >> 8])
    if mc >= 15:
        v = mc - 15
        while v >= 255: out.append(255); v -= 255
        out.append(v)
    return out


def _lz4_tail(lit):
    out = bytearray()
    ll = len(lit)
    out.append(min(ll, 15) << 4)
    if ll >= 15:
        v = ll - 15
        while v >= 255: out.append(255); v -= 255
        out.append(v)
    out += bytes(lit)
    return out


def test_handbuilt_lz4_streams(pkg, oracle):
    """Adversarial LZ4 blocks: overlapping matches at every small offset and at the copy-path
    boundaries (63/64/65, 1023/1024/1025), matches that read the literals of their own sequence,
    long literal runs, long runs — the cases the wave-cooperative copy logic branches on."""
    rng = np.random.default_rng(11)
    streams = []
    for off in list(range(1, 70)) + [127, 128, 129, 255, 256, 257, 1000, 1023, 1024, 1025, 2047, 2048, 4099]:
        for mlen in [4, 5, 7, 8, 15, 16, 18, 19, 31, 63, 64, 65, 100, 255, 256, 300, 1023, 1024, 1025, 2047, 2048, 2049, 5000, 70000]:
            s = bytearray()
            pre = rng.integers(0, 256, max(off, 16) + 3, dtype=np.uint8).tobytes()
            s += _lz4_seq(pre, off, mlen)
            # a second sequence whose match source overlaps the literals it has just emitted
            lit2 = rng.integers(0, 256, 5, dtype=np.uint8).tobytes()
            s += _lz4_seq(lit2, 3, 9)
            s += _lz4_seq(b"", 1, 40)
            s += _lz4_tail(rng.integers(0, 256, 12, dtype=np.uint8).tobytes())
            streams.append(bytes(s))
    # long literal runs around the register-window limits
    for ll in [0, 1, 14, 15, 16, 63, 64, 65, 200, 255, 256, 270, 300, 500, 511, 512, 513, 1024, 5000, 70000]:
        s = bytearray()
        s += _lz4_seq(rng.integers(0, 256, ll + 4, dtype=np.uint8).tobytes(), 4, 20)
        s += _lz4_seq(rng.integers(0, 256, ll, dtype=np.uint8).tobytes(), 7, 4)
        s += _lz4_tail(rng.integers(0, 256, ll + 13, dtype=np.uint8).tobytes())
        streams.append(bytes(s))
    bad = []
    for i, s in enumerate(streams):
        s = np.frombuffer(s, np.uint8)
        cap = 1 << 20
        tmp = np.zeros(cap, np.uint8)
        n = oracle.orc_lz4_decompress(ptr(s), s.size, ptr(tmp), cap)
        # exact-size decode (blosc requires the codec to produce exactly neblock bytes)
        want = np.zeros(n, np.uint8)
        assert oracle.orc_lz4_decompress(ptr(s), s.size, ptr(want), n) == n
        chunk = wrap_stream_as_chunk(s, n, 1)
        r, out = pkg.decompress(chunk, n)
        if r != n or not np.array_equal(out, want):
            bad.append((i, r, n))
    assert not bad, bad[:20]


# ---- hand-built BloscLZ streams (grammar of blosclz.c:679-789) --------------------------------------------
def _blz_lits(b):
    """literal runs of at most 32 bytes: control byte n-1, then the bytes"""
    out = bytearray()
    b = bytes(b)
    for k in range(0, len(b), 32):
        run = b[k:k + 32]
        out.append(len(run) - 1)
        out += run
    return out


def _blz_match(dist, length):
    """one match: total length = (ctrl>>5) + 2 (+ extension bytes when the field is 7), distance-1 in 13 bits, or
    the escape 31/255 followed by a big-endian 16-bit value + 8191 (blosclz.c:700-733)"""
    assert length >= 3 and 1 <= dist <= 65535 + 8191 + 1
    out = bytearray()
    d = dist - 1
    far = d >= 8191
    hi = 31 if far else d >> 8
    if length <= 8:
        out.append(((length - 2) << 5) | hi)
    else:
        out.append((7 << 5) | hi)
        v = length - 9
        while v >= 255:
            out.append(255); v -= 255
        out.append(v)
    if far:
        out.append(255)
        out += bytes([(d - 8191) >> 8, (d - 8191) & 255])
    else:
        out.append(d & 255)
    return out


def test_handbuilt_blosclz_streams(pkg, oracle):
    """Adversarial BloscLZ streams for the batched step and the scalar path of blosclz_decode_wave: distance-1
    runs, overlapping matches at every small distance, the 8191/8192 boundary of the far-distance escape, length
    fields 3..9 and every extension-chain shape (9, 263, 264, 265, 519 ...), literal runs of 1..32, a match that
    reads the literals just emitted, and the 'pending match dropped at end of input' quirk (blosclz.c:688-736)."""
    rng = np.random.default_rng(12)
    streams = []
    dists = list(range(1, 70)) + [127, 128, 255, 256, 257, 1023, 1024, 4096, 8190, 8191, 8192, 8193, 9000, 20000, 65535, 65535 + 8191 + 1]
    lens = [3, 4, 5, 8, 9, 10, 16, 17, 31, 63, 64, 65, 100, 263, 264, 265, 300, 519, 1024, 1025, 2049, 5000, 70000]
    for dist in dists:
        for mlen in lens:
            s = bytearray()
            pre = rng.integers(0, 256, max(dist, 4) + int(rng.integers(0, 5)), dtype=np.uint8).tobytes()
            s += _blz_lits(pre)
            s += _blz_match(dist, mlen)
            s += _blz_lits(rng.integers(0, 256, int(rng.integers(1, 33)), dtype=np.uint8).tobytes())
            s += _blz_match(3, 9)            # reads the literals just emitted
            s += _blz_match(1, 40)           # distance-1 run
            s += _blz_lits(rng.integers(0, 256, 7, dtype=np.uint8).tobytes())
            streams.append(bytes(s))
    # dense short tokens: what the batched step sees on real data
    for trial in range(60):
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
    # the end-of-input quirk: a stream that ENDS with a match (no byte behind it) drops that match
    for dist, mlen in [(1, 3), (5, 9), (300, 264), (9000, 20)]:
        s = bytearray(_blz_lits(rng.integers(0, 256, max(dist, 16), dtype=np.uint8).tobytes()))
        s += _blz_match(2, 10)
        s += _blz_match(dist, mlen)
        streams.append(bytes(s))
    bad = []
    for i, s in enumerate(streams):
        s = np.frombuffer(s, np.uint8)
        cap = 1 << 20
        tmp = np.zeros(cap, np.uint8)
        n = oracle.orc_blosclz_decompress(ptr(s), s.size, ptr(tmp), cap)
        if n <= 0:       # e.g. a 2-byte match token as the stream's last bytes (blosclz.c:707-710): rejected by both
            assert i >= len(streams) - 4, i
            ch = wrap_stream_as_chunk(s, 4096, 0)
            r0, _ = pkg.decompress(ch, 4096)
            ro, _ = orc_decompress(oracle, ch, 4096)
            if not (r0 < 0 and ro < 0):
                bad.append((i, "rejected stream", r0, ro))
            continue
        # chunk level (blosc_d hands the codec exactly neblock bytes of room, blosc.c:777-782): verdict and bytes of the oracle
        chunk = wrap_stream_as_chunk(s, n, 0)
        r, out = pkg.decompress(chunk, n)
        ro, oo = orc_decompress(oracle, chunk, n)
        if i < len(streams) - 4:
            assert ro == n and np.array_equal(oo, tmp[:n]), i
        if r != ro or (ro == n and not np.array_equal(out, oo)):
            bad.append((i, r, ro, n))
        # one byte short / one byte long: blosc_d requires exactly neblock bytes (blosc.c:780-782)
        for wrong in (n - 1, n + 1):
            if wrong <= 0:
                continue
            r2, _ = pkg.decompress(wrap_stream_as_chunk(s, wrong, 0), wrong)
            ro, _ = orc_decompress(oracle, wrap_stream_as_chunk(s, wrong, 0), wrong)
            if (r2 < 0) != (ro < 0):
                bad.append((i, "size", wrong, r2, ro))
    assert not bad, bad[:20]


def test_handbuilt_dense_near_match_chains(pkg, oracle):
    """LZ4 blocks made of MANY short sequences per 64 stream bytes whose matches reach back a few bytes - into the output of
    the sequences right before them: what the LDS-assembled step of the decoder (k_decode.hip: lz4_step_lds) exists for.
    Random chains with distances 1..40 (overlapping and not), 0..3 literals in between, matches that read the literals of
    their own step, chains right at the start of the block (less history than the step asks for), chains whose history
    need straddles the 1 KiB limit, long matches mixed in (the step must fall back), steps around the 2 KiB output limit."""
    rng = np.random.default_rng(77)

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

    streams = []
    for first_lit in (1, 2, 5, 17, 64, 300, 1030, 1100):               # start of block: history shorter than asked for / around 1 KiB
        for maxoff in (1, 2, 3, 7, 16, 40, 64, 700, 1024, 1030, 1500):
            for maxml, maxll in ((4, 0), (8, 1), (18, 3), (30, 2), (64, 0), (140, 1), (273, 0)):
                streams.append(chain(int(rng.integers(20, 400)), maxoff, maxml, maxll, first_lit))
    bad = []
    for i, s in enumerate(streams):
        s = np.frombuffer(s, np.uint8)
        cap = 1 << 20
        tmp = np.zeros(cap, np.uint8)
        n = oracle.orc_lz4_decompress(ptr(s), s.size, ptr(tmp), cap)
        assert n > 0
        want = tmp[:n].copy()
        chunk = wrap_stream_as_chunk(s, n, 1)
        r, out = pkg.decompress(chunk, n)
        if r != n or not np.array_equal(out, want):
            bad.append((i, r, n, int(np.argmax(out[:n] != want)) if r == n else -1))
    assert not bad, bad[:10]
    # the same chains as the T planes of split shuffled blocks (the fused path), 8 streams of equal size per chunk
    T = 8
    groups = {}
    for s in streams[:200]:
        s = np.frombuffer(s, np.uint8)
        tmp = np.zeros(1 << 20, np.uint8)
        n = oracle.orc_lz4_decompress(ptr(s), s.size, ptr(tmp), 1 << 20)
        groups.setdefault(n, []).append((s, tmp[:n].copy()))
    checked = 0
    for n, lst in groups.items():
        if n < 128:
            continue
        planes = (lst * T)[:T]
        chunk = wrap_planes_as_chunk([bytes(p[0]) for p in planes], n, 1)
        want = np.stack([p[1] for p in planes], 1).reshape(-1)
        r, out = pkg.decompress(chunk, want.size)
        assert r == want.size and np.array_equal(out, want), ("planes", n, r)
        checked += 1
    assert checked > 20
