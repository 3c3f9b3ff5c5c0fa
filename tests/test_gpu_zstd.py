"""GPU: Zstd decode (codec row K8, decode direction; c-blosc_amd/csrc/k_zstd.hip) through the C ABI against the
oracle: the committed chunks written by the real reference (clevel 1-9, typesize 1-8, shuffle / bitshuffle /
none, forced block size), getitem on them, the batched device-resident call, and corrupted chunks (same
verdict as the oracle, never a hang)."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import DATASETS, orc_decompress, ptr

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _chunks():
    z = np.load(os.path.join(GOLDEN, "ref_zstd_chunks.npz"))
    for k, m in enumerate(z["meta"]):
        dname, n, T, clevel, shuffle, bs = m.split(",")
        yield z[f"c{k}"], dname, int(n), int(T)


def test_reference_written_zstd_chunks(pkg):
    for chunk, dname, n, T in _chunks():
        r, out = pkg.decompress(chunk, n)
        assert r == n and np.array_equal(out, DATASETS[dname](n)), (dname, n, T)


def test_getitem_on_zstd_chunks(pkg, lib):
    for chunk, dname, n, T in _chunks():
        data = DATASETS[dname](n)
        nel = n // T
        for start, nitems in ((0, 1), (nel // 3, min(5000, nel - nel // 3)), (nel - 7, 7)):
            got = np.zeros(nitems * T, np.uint8)
            assert lib.blosc_getitem(ptr(chunk), start, nitems, ptr(got)) == nitems * T
            assert np.array_equal(got, data[start * T:(start + nitems) * T]), (dname, start, nitems)


def test_device_batch_mixed_codecs(pkg, oracle):
    """one batched call over Zstd, zlib, LZ4 and BloscLZ chunks together (three kernels, three sets of queues over one block table:
    k_decode_streams' per-XCD queues leave the Zstd / zlib blocks out, the zlib kernel's hold exactly its own - engine.hip, round 3)"""
    import torch
    from helpers import orc_compress
    dev = torch.device("cuda:0")
    items = [(c, DATASETS[d](n)) for c, d, n, T in list(_chunks())[:8]]
    zl = np.load(os.path.join(GOLDEN, "ref_zlib_chunks.npz"))
    for k in (0, 1, 4, 10, 15, 20):
        dname, n, T, clevel, shuffle, bs = zl["meta"][k].split(",")
        items.insert(2 * (k % 4), (zl[f"c{k}"], DATASETS[dname](int(n))))            # interleaved with the Zstd chunks
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


def test_corrupt_zstd_chunks_same_verdict_as_oracle(pkg, oracle, ref):
    """damaged reference-written chunks: the GPU decoder against the oracle (same verdict, same bytes) AND against the real
    reference where oracle/_ref ships: whatever stock blosc_decompress_ctx rejects is rejected, whatever both accept has the
    reference's bytes.  (The reference's default 64-bit build also accepts frames with a Huffman literal stream that is not
    consumed exactly - its fast loop skips that check, huf_decompress.c:873-887 vs :692-693 - where oracle and GPU decoder
    follow RFC 8878 4.2.2 and its portable loop: tests/test_oracle_zstd.py pins that as the only difference.)"""
    from helpers import ref_decompress
    rng = np.random.default_rng(21)
    chunks = list(_chunks())
    nref_acc = 0
    for trial in range(240):
        chunk, dname, n, T = chunks[trial % min(len(chunks), 6)]
        c = chunk.copy()
        pos = int(rng.integers(16, c.size))
        if trial % 4 == 3: c[pos] = int(rng.integers(0, 256))
        else: c[pos] ^= 1 << int(rng.integers(0, 8))
        ro, oo = orc_decompress(oracle, c, n)
        rg, og = pkg.decompress(c, n)
        if ro == n:
            assert rg == n and np.array_equal(og, oo), (trial, pos)
        else:
            assert rg < 0, (trial, pos, ro, rg)
        if ref is not None:
            rr, want = ref_decompress(ref, c, n)
            if rr != n:
                assert rg < 0, (trial, pos, rr, rg)
            elif rg == n:
                assert np.array_equal(og, want), (trial, pos)
                nref_acc += 1
    assert ref is None or nref_acc >= 5


def test_content_checksum_frames(pkg, oracle, ref):
    """frames that carry a Content_Checksum (blosc never writes them, a caller's pre-made chunk may): decoded when the
    checksum fits, rejected (-1) when it does not - the reference's verdicts."""
    if ref is None:
        pytest.skip("needs oracle/_ref to write frames with a checksum")
    from test_oracle_zstd import _checksum_frames
    for data, frame in _checksum_frames(ref):
        n = data.size
        if n < 128:
            continue
        for damage in (False, True):
            f = frame.copy()
            if damage:
                f[-2] ^= 0x10
            total = 16 + 4 + 4 + f.size                      # one unsplit block: header, bstarts[1], csize, frame
            c = np.zeros(total, np.uint8)
            c[0] = 2; c[1] = 1; c[2] = 0x10 | (4 << 5); c[3] = 1
            c[4:8] = np.array([n], "<i4").view(np.uint8); c[8:12] = np.array([n], "<i4").view(np.uint8)
            c[12:16] = np.array([total], "<i4").view(np.uint8)
            c[16:20] = np.array([20], "<i4").view(np.uint8); c[20:24] = np.array([f.size], "<i4").view(np.uint8)
            c[24:] = f
            ro, oo = orc_decompress(oracle, c, n)
            rg, og = pkg.decompress(c, n)
            if damage:
                assert ro < 0 and rg < 0, (n, ro, rg)
            else:
                assert ro == n and rg == n and np.array_equal(og, data), (n, ro, rg)
