"""GPU: reference-written chunks of EVERY data set of SURVEY 8d at the bench's chunk size, across typesizes, both filters and both LZ codecs, every
byte of every chunk compared - and the same inputs compressed here and read by the reference.  Round 5 found a silent decode error (right sizes, wrong
bytes) on one cell of this grid that no other test visited: the reference's `linspace` chunks at typesize 4 (tests/test_gpu_spans.py has the
mechanism).  The benchmark's own cells are in tests/test_gpu_baseline_geometry.py; this is the grid around them."""
import numpy as np
import pytest

from helpers import DATASETS, orc_compress, ref_compress, ref_decompress, orc_decompress

pytestmark = pytest.mark.gpu
CSZ = 64 << 20
NCH = 4


@pytest.mark.parametrize("dname", list(DATASETS))
def test_grid_around_the_bench_cells(pkg, lib, oracle, ref, dname):
    import torch
    dev = torch.device("cuda:0")
    data = DATASETS[dname](CSZ)
    d_data = torch.from_numpy(data).to(dev)
    src = d_data.unsqueeze(0).expand(NCH, CSZ).contiguous()
    comp = torch.zeros((NCH, CSZ + 256), dtype=torch.uint8, device=dev)
    back = torch.zeros((NCH, CSZ), dtype=torch.uint8, device=dev)
    bad = []
    for T in (1, 2, 4, 8, 16, 3):
        for shuffle in (1, 2, 0):
            if shuffle == 1 and T == 1:
                continue
            for codec in ("lz4", "blosclz"):
                for clevel in ((5, 9, 1) if T in (4, 8) and shuffle else (5,)):
                    if ref is not None:
                        r, stock = ref_compress(ref, data, T, clevel, shuffle, codec.encode(), nthreads=16)
                    else:
                        r, stock = orc_compress(oracle, data, T, clevel, shuffle, codec)
                    assert r > 0
                    # ---- reference-written chunks, every byte of the batch ----
                    comp[:, :stock.size].copy_(torch.from_numpy(stock).to(dev).unsqueeze(0).expand(NCH, stock.size))
                    back.fill_(0xEE)
                    bd = pkg.DeviceBatch([comp[i].data_ptr() for i in range(NCH)], [CSZ + 16] * NCH, [back[i].data_ptr() for i in range(NCH)], [CSZ] * NCH)
                    for rep in range(2):      # (the second call runs in the cost-feedback order)
                        ok = bd.decompress() == 0 and bd.results() == [CSZ] * NCH and bool(torch.equal(back, src))
                        if not ok:
                            bad.append(("decode of reference-written chunks", T, shuffle, codec, clevel, rep, int((back != src).sum())))
                    # ---- written here, read by the reference (first and last chunk) and by this library ----
                    bc = pkg.DeviceBatch([src[i].data_ptr() for i in range(NCH)], [CSZ] * NCH, [comp[i].data_ptr() for i in range(NCH)], [CSZ + 16] * NCH)
                    if bc.compress(T, clevel, shuffle, codec.encode()) != 0 or min(bc.results()) <= 0:
                        bad.append(("compress", T, shuffle, codec, clevel, bc.results()[:2])); continue
                    cb = bc.results()
                    for i in (0, NCH - 1):
                        ch = comp[i][:cb[i]].cpu().numpy()
                        rr, out = ref_decompress(ref, ch, CSZ) if ref is not None else orc_decompress(oracle, ch, CSZ)
                        if rr != CSZ or not np.array_equal(out, data):
                            bad.append(("the reference reading a chunk written here", T, shuffle, codec, clevel, i))
                    back.fill_(0xEE)
                    bd2 = pkg.DeviceBatch([comp[i].data_ptr() for i in range(NCH)], [CSZ + 16] * NCH, [back[i].data_ptr() for i in range(NCH)], [CSZ] * NCH)
                    if not (bd2.decompress() == 0 and bd2.results() == [CSZ] * NCH and bool(torch.equal(back, src))):
                        bad.append(("decode of own chunks", T, shuffle, codec, clevel, int((back != src).sum())))
    assert not bad, bad[:12]


@pytest.mark.parametrize("dname", list(DATASETS))
def test_grid_of_the_entropy_coded_formats(pkg, lib, oracle, ref, dname):
    """The same grid for the formats only the real reference can write (Zstd, Zlib, LZ4HC): reference-written chunks decoded here byte for byte, chunks
    written here decoded by the reference."""
    if ref is None:
        pytest.skip("oracle/_ref is not built: nothing writes Zstd / Zlib / LZ4HC chunks")
    import torch
    dev = torch.device("cuda:0")
    nch = 2
    data = DATASETS[dname](CSZ)
    d_data = torch.from_numpy(data).to(dev)
    src = d_data.unsqueeze(0).expand(nch, CSZ).contiguous()
    comp = torch.zeros((nch, CSZ + 256), dtype=torch.uint8, device=dev)
    back = torch.zeros((nch, CSZ), dtype=torch.uint8, device=dev)
    bad = []
    for codec, clevel in (("zstd", 3), ("zlib", 5), ("lz4hc", 9), ("zstd", 7)):
        for T in (4, 8, 2):
            for shuffle in (1, 2):
                r, stock = ref_compress(ref, data, T, clevel, shuffle, codec.encode(), nthreads=16)
                assert r > 0
                comp[:, :stock.size].copy_(torch.from_numpy(stock).to(dev).unsqueeze(0).expand(nch, stock.size))
                back.fill_(0xEE)
                bd = pkg.DeviceBatch([comp[i].data_ptr() for i in range(nch)], [CSZ + 16] * nch, [back[i].data_ptr() for i in range(nch)], [CSZ] * nch)
                if not (bd.decompress() == 0 and bd.results() == [CSZ] * nch and bool(torch.equal(back, src))):
                    bad.append(("decode of reference-written chunks", T, shuffle, codec, clevel, int((back != src).sum())))
                bc = pkg.DeviceBatch([src[i].data_ptr() for i in range(nch)], [CSZ] * nch, [comp[i].data_ptr() for i in range(nch)], [CSZ + 16] * nch)
                if bc.compress(T, clevel, shuffle, codec.encode()) != 0 or min(bc.results()) <= 0:
                    bad.append(("compress", T, shuffle, codec, clevel, bc.results()[:2])); continue
                cb = bc.results()
                ch = comp[nch - 1][:cb[nch - 1]].cpu().numpy()
                rr, out = ref_decompress(ref, ch, CSZ)
                if rr != CSZ or not np.array_equal(out, data):
                    bad.append(("the reference reading a chunk written here", T, shuffle, codec, clevel))
                back.fill_(0xEE)
                bd2 = pkg.DeviceBatch([comp[i].data_ptr() for i in range(nch)], [CSZ + 16] * nch, [back[i].data_ptr() for i in range(nch)], [CSZ] * nch)
                if not (bd2.decompress() == 0 and bd2.results() == [CSZ] * nch and bool(torch.equal(back, src))):
                    bad.append(("decode of own chunks", T, shuffle, codec, clevel, int((back != src).sum())))
    assert not bad, bad[:12]


@pytest.mark.parametrize("dname", ["linspace", "bench19", "randwalk", "smallints"])
def test_ragged_chunk_sizes_and_forced_blocks_at_scale(pkg, lib, oracle, ref, dname):
    """Chunks that do not end on a block boundary (a leftover block that is one unsplit stream), that are not a multiple of the typesize, with forced
    block sizes and the other split modes - at tens of MiB, where the queues, spans and hand-offs are the benchmark's; host buffers through the
    stock entry points (one chunk per call)."""
    full = DATASETS[dname](48 << 20)
    writer = (lambda d, T, cl, sh, c, bs: ref_compress(ref, d, T, cl, sh, c.encode(), blocksize=bs, nthreads=16)) if ref is not None else \
             (lambda d, T, cl, sh, c, bs: orc_compress(oracle, d, T, cl, sh, c, blocksize=bs))
    bad = []
    for n, T, shuffle, codec, bs in [((40 << 20) - 12345, 8, 1, "lz4", 0), ((33 << 20) + 7, 4, 1, "lz4", 0), ((33 << 20) + 7, 4, 2, "blosclz", 0),
                                     ((24 << 20) + 1000, 8, 1, "blosclz", 65536), ((24 << 20) + 1000, 2, 1, "lz4", 40000), ((20 << 20) + 24, 8, 2, "lz4", 1 << 20),
                                     ((16 << 20) + 3, 16, 1, "lz4", 0), ((16 << 20) + 3, 3, 1, "lz4", 0), ((16 << 20) - 1, 1, 2, "lz4", 0)]:
        data = full[:n]
        r, stock = writer(data, T, 5, shuffle, codec, bs)
        assert r > 0
        rr, got = pkg.decompress(stock, n)
        if rr != n or not np.array_equal(got, data):
            bad.append(("decode of a reference-written chunk", n, T, shuffle, codec, bs, int((got != data).sum()) if rr == n else rr))
        rc, chunk = pkg.compress(data, T, 5, shuffle, codec.encode(), blocksize=bs)
        if rc <= 0:
            bad.append(("compress", n, T, shuffle, codec, bs, rc)); continue
        ro, out = ref_decompress(ref, chunk, n) if ref is not None else orc_decompress(oracle, chunk, n)
        if ro != n or not np.array_equal(out, data):
            bad.append(("the reference reading a chunk written here", n, T, shuffle, codec, bs))
    assert not bad, bad
