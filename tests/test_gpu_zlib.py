"""GPU: Zlib decode (codec row "Zlib" of SURVEY §8f-3; c-blosc_amd/csrc/k_zlib.hip) through the C ABI: the five compat vectors
of the reference, the committed chunks written by the real reference (clevel 1-9, typesize 1-8, shuffle / bitshuffle / none,
forced block size), getitem on them, the batched device-resident call, hand-built legal and illegal zlib streams wrapped
into chunks (tests/deflate_builder.py), and corrupted chunks - verdict AND bytes compared with the oracle and, where
oracle/_ref travels along, with the reference itself."""
import glob
import os
import zlib

import numpy as np
import pytest

import deflate_builder as D
from helpers import DATASETS, orc_decompress, ptr, ref_decompress
from test_oracle_zlib import HANDBUILT

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _chunks():
    z = np.load(os.path.join(GOLDEN, "ref_zlib_chunks.npz"))
    for k, m in enumerate(z["meta"]):
        dname, n, T, clevel, shuffle, bs = m.split(",")
        yield z[f"c{k}"], dname, int(n), int(T)


def one_stream_chunk(stream, n, typesize=1):
    """an unsplit single-block Zlib chunk around `stream` (header, bstarts[1], csize, stream): blosc.c:1148-1247 layout"""
    total = 16 + 4 + 4 + stream.size
    c = np.zeros(total, np.uint8)
    c[0] = 2; c[1] = 1; c[2] = 0x10 | (3 << 5); c[3] = typesize          # dont_split, codec format 3 (Zlib), no filter
    c[4:8] = np.array([n], "<i4").view(np.uint8); c[8:12] = np.array([n], "<i4").view(np.uint8)
    c[12:16] = np.array([total], "<i4").view(np.uint8)
    c[16:20] = np.array([20], "<i4").view(np.uint8); c[20:24] = np.array([stream.size], "<i4").view(np.uint8)
    c[24:] = stream
    return c


@pytest.mark.parametrize("fname", sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "compat", "*zlib*.cdata"))))
def test_compat_zlib_vectors(pkg, fname):
    chunk = np.fromfile(os.path.join(GOLDEN, "compat", fname), np.uint8)
    r, out = pkg.decompress(chunk, 4000000)
    assert r == 4000000 and np.array_equal(out.view("<i4"), np.arange(10**6, dtype="<i4"))


def test_reference_written_zlib_chunks(pkg):
    for chunk, dname, n, T in _chunks():
        r, out = pkg.decompress(chunk, n)
        assert r == n and np.array_equal(out, DATASETS[dname](n)), (dname, n, T)


def test_getitem_on_zlib_chunks(pkg, lib):
    for chunk, dname, n, T in _chunks():
        data = DATASETS[dname](n)
        nel = n // T
        for start, nitems in ((0, 1), (nel // 3, min(5000, nel - nel // 3)), (nel - 7, 7)):
            got = np.zeros(nitems * T, np.uint8)
            assert lib.blosc_getitem(ptr(chunk), start, nitems, ptr(got)) == nitems * T
            assert np.array_equal(got, data[start * T:(start + nitems) * T]), (dname, start, nitems)


def test_device_batch_mixed_codecs(pkg, oracle):
    """one batched call over Zlib, LZ4 and BloscLZ chunks together"""
    import torch
    from helpers import orc_compress
    dev = torch.device("cuda:0")
    items = [(c, DATASETS[d](n)) for c, d, n, T in list(_chunks())[:8]]
    for codec in ("lz4", "blosclz"):
        data = DATASETS["bench19"](1 << 20)
        r, ch = orc_compress(oracle, data, 8, 5, 1, codec)
        items.append((ch[:r].copy(), data))
    d_src = [torch.from_numpy(c.copy()).to(dev) for c, _ in items]
    d_dst = [torch.zeros(p.size, dtype=torch.uint8, device=dev) for _, p in items]
    b = pkg.DeviceBatch([t.data_ptr() for t in d_src], [c.size for c, _ in items], [t.data_ptr() for t in d_dst], [p.size for _, p in items])
    assert b.decompress() == 0
    assert b.results() == [p.size for _, p in items]
    for t, (_, p) in zip(d_dst, items):
        assert np.array_equal(t.cpu().numpy(), p)


def test_stock_zlib_streams_every_level_and_strategy(pkg, oracle):
    """streams written by zlib itself: levels 0-9 (stored blocks at 0), fixed / Huffman-only / RLE strategies, small windows,
    several blocks - wrapped into one-block chunks"""
    for name in ("bench19", "linspace", "randwalk", "zeros", "smallints"):
        for n in (130, 4096, 70000, 300000):
            d = DATASETS[name](n); b = d.tobytes()
            streams = [zlib.compress(b, lvl) for lvl in (0, 1, 6, 9)]
            for strat, wb in ((zlib.Z_FIXED, 15), (zlib.Z_HUFFMAN_ONLY, 15), (zlib.Z_RLE, 9)):
                co = zlib.compressobj(6, zlib.DEFLATED, wb, 8, strat); streams.append(co.compress(b) + co.flush())
            co = zlib.compressobj(5); step = max(1, len(b) // 5); parts = []
            for k in range(0, len(b), step):
                parts += [co.compress(b[k:k + step]), co.flush(zlib.Z_FULL_FLUSH)]
            streams.append(b"".join(parts) + co.flush())
            for s in streams:
                s = np.frombuffer(s, np.uint8)
                if s.size >= n:
                    continue                      # a chunk cannot hold a split that did not shrink (it would be stored raw)
                c = one_stream_chunk(s, n)
                ro, oo = orc_decompress(oracle, c, n)
                rg, og = pkg.decompress(c, n)
                assert ro == n and rg == n and np.array_equal(og, d), (name, n, s.size, ro, rg)


def test_handbuilt_streams_same_verdict_as_reference(pkg, oracle, ref):
    """legal oddities and illegal blocks: accepted exactly when the reference's zlib produces exactly n bytes"""
    for name, s, cap in D.cases():
        good = HANDBUILT[name]
        for n in sorted({max(good, 128), 128, 200}):
            if s.size >= n:
                continue
            c = one_stream_chunk(s, n)
            ro, oo = orc_decompress(oracle, c, n)
            rg, og = pkg.decompress(c, n)
            expect_ok = good == n
            assert (ro == n) == expect_ok, (name, n, ro)
            assert (rg == n) == expect_ok and (rg == n or rg < 0), (name, n, rg)
            if expect_ok:
                assert np.array_equal(og, oo)
            if ref is not None:
                rr, orr = ref_decompress(ref, c, n)
                assert (rr == n) == expect_ok, (name, n, rr)


def test_corrupt_zlib_chunks_same_verdict_as_oracle_and_reference(pkg, oracle, ref):
    rng = np.random.default_rng(23)
    for chunk, dname, n, T in list(_chunks())[:3]:
        for trial in range(40):
            c = chunk.copy()
            pos = int(rng.integers(16, c.size)); c[pos] ^= 1 << int(rng.integers(0, 8))
            ro, oo = orc_decompress(oracle, c, n)
            rg, og = pkg.decompress(c, n)
            if ro == n:
                assert rg == n and np.array_equal(og, oo), (trial, pos)
            else:
                assert rg < 0, (trial, pos, ro, rg)
            if ref is not None:
                rr, _ = ref_decompress(ref, c, n)
                assert (rr == n) == (rg == n), (trial, pos, rr, rg)


@pytest.mark.parametrize("fixture", ["ref_zlib_chunks.npz", "ref_zstd_chunks.npz"])
def test_corrupt_chunks_in_canary_padded_device_buffers(pkg, oracle, fixture):
    """Random damage to reference-written Zlib / Zstd chunks, device-resident, output in canary-padded buffers: the verdict
    equals the oracle's, accepted chunks carry the oracle's bytes, and nothing is written outside [dest, dest + nbytes) -
    the literal scatter and the batched match execution of the two entropy-coded decoders included."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(41)
    PAD = 4096
    z = np.load(os.path.join(GOLDEN, fixture))
    metas = [m.split(",") for m in z["meta"]]
    for k in (0, 1, 2, 5, 9):
        good = z[f"c{k}"]; n = int(metas[k][1])
        nblocks_hdr = 16 + 4 * ((n + int(good[8:12].view("<i4")[0]) - 1) // int(good[8:12].view("<i4")[0]))
        chunks, want = [], []
        for trial in range(24):
            c = good.copy()
            cnt = int(rng.integers(1, 6))
            pos = rng.integers(nblocks_hdr, c.size, cnt)
            c[pos] = rng.integers(0, 256, cnt, dtype=np.uint8)
            ro, oo = orc_decompress(oracle, c, n)
            want.append((ro, oo.copy())); chunks.append(c)
        d_src = [torch.from_numpy(c).to(dev) for c in chunks]
        d_dst = [torch.full((n + 2 * PAD,), 0xA5, dtype=torch.uint8, device=dev) for _ in chunks]
        b = pkg.DeviceBatch([t.data_ptr() for t in d_src], [c.size for c in chunks], [t.data_ptr() + PAD for t in d_dst], [n] * len(chunks))
        assert b.decompress() == 0
        for i, (rg, (ro, oo)) in enumerate(zip(b.results(), want)):
            out = d_dst[i].cpu().numpy()
            assert (out[:PAD] == 0xA5).all() and (out[PAD + n:] == 0xA5).all(), (fixture, k, i, "canary")
            if ro == n:
                assert rg == n and np.array_equal(out[PAD:PAD + n], oo), (fixture, k, i, rg)
            else:
                assert rg < 0, (fixture, k, i, rg, ro)
        r, out = pkg.decompress(good, n)
        assert r == n
