"""GPU: the BASELINE.json geometry as a parity test, not only as a benchmark: 64 MiB chunks, a batch of
128 = the bench's full 8 GiB for BASELINE.json's own configurations (#2, #3, #4, #1's call on the GPU: FULL below), of 64 (4 GiB) for
the other cases - >= 4096 streams either way, so the per-XCD queues, the cost-feedback order and the look-ahead of the persistent
kernels are the ones bench.py runs.  Decompression is checked on chunks written by the reference itself
(oracle/_ref when it ships, else the oracle) in every chunk of the batch; compression by letting the reference
decode GPU-written chunks from the batch's ends and middle."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import DATASETS, header, orc_compress, orc_decompress, ref_compress, ref_decompress

pytestmark = pytest.mark.gpu
CSZ = 64 << 20
FULL = {"lz4-shuffle-T8", "lz4-bitshuffle-T4", "blosclz-shuffle-T8", "zstd-shuffle-T8"}      # 128 chunks: exactly what bench.py times

CASES = [  # name, codec, shuffle, typesize, clevel, dataset, GPU encodes it
    ("lz4-shuffle-T8", "lz4", 1, 8, 5, "bench19", True),
    ("lz4-bitshuffle-T4", "lz4", 2, 4, 5, "bench19", True),
    ("blosclz-shuffle-T8", "blosclz", 1, 8, 5, "bench19", True),
    ("lz4-bitshuffle-T8", "lz4", 2, 8, 5, "bench19", True),      # round 5: bit(un)shuffle of typesize 8 inside the codec kernels
    ("lz4-shuffle-T8-linspace", "lz4", 1, 8, 5, "linspace", True),
    ("lz4-shuffle-T8-randwalk", "lz4", 1, 8, 5, "randwalk", True),
    ("zstd-shuffle-T8", "zstd", 1, 8, 3, "bench19", True),
    ("zlib-shuffle-T8", "zlib", 1, 8, 5, "bench19", True),
    # typesize 2 and 16: their byte (un)shuffle runs inside the codec kernels since round 3 (lane-table spans, quad transposes: DESIGN.md 3.2),
    # for the entropy-coded formats by the decoding wave (Zstd: unsplit blocks) or through the zlib kernel's per-XCD hand-off
    ("lz4-shuffle-T2", "lz4", 1, 2, 5, "bench19", True),
    ("lz4-shuffle-T16", "lz4", 1, 16, 5, "bench19", True),
    ("blosclz-shuffle-T16-linspace", "blosclz", 1, 16, 5, "linspace", True),
    ("zstd-shuffle-T16", "zstd", 1, 16, 3, "bench19", True),
    ("zlib-shuffle-T2", "zlib", 1, 2, 5, "linspace", True),
    # the encoder options that are not the default at these settings: the LZ4HC-grade search in front of the Zstd
    # writer (the default from clevel 6 on), Huffman-coded literals, and the round-2 forms of both writers
    ("zstd-cl7-search", "zstd", 1, 8, 7, "bench19", True),
    ("zstd-huffman-smallints", "zstd", 1, 4, 3, "smallints", True, {"BLOSC_AMD_ZSTD_HUFFMAN": "1"}),
    ("zstd-predefined-tables", "zstd", 1, 8, 3, "bench19", True, {"BLOSC_AMD_ZSTD_TABLES": "0"}),
    ("zlib-fixed-codes", "zlib", 1, 8, 5, "bench19", True, {"BLOSC_AMD_ZLIB_DYNAMIC": "0", "BLOSC_AMD_ZLIB_SEARCH": "0"}),
    ("zlib-search-only", "zlib", 1, 8, 5, "linspace", True, {"BLOSC_AMD_ZLIB_DYNAMIC": "0", "BLOSC_AMD_ZLIB_SEARCH": "1"}),
]
CASES = [c if len(c) == 8 else c + ({},) for c in CASES]


@pytest.mark.parametrize("name,codec,shuffle,T,clevel,dname,gpu_enc,env", CASES, ids=[c[0] for c in CASES])
def test_batch_at_baseline_geometry(pkg, lib, oracle, ref, monkeypatch, name, codec, shuffle, T, clevel, dname, gpu_enc, env):
    import torch
    NCHUNKS = 128 if name in FULL else 64
    for k, v in env.items():          # the encoder switches are read per call (engine.hip)
        monkeypatch.setenv(k, v)
    dev = torch.device("cuda:0")
    data = DATASETS[dname](CSZ)
    if ref is not None:
        r, stock = ref_compress(ref, data, T, clevel, shuffle, codec.encode(), nthreads=8)
    else:
        if codec in ("zstd", "zlib"):
            pytest.skip("no Zstd / Zlib writer without oracle/_ref")
        r, stock = orc_compress(oracle, data, T, clevel, shuffle, codec)
    assert r > 0
    d_data = torch.from_numpy(data).to(dev)
    src = d_data.unsqueeze(0).expand(NCHUNKS, CSZ).contiguous()
    comp = torch.zeros((NCHUNKS, CSZ + 256), dtype=torch.uint8, device=dev)
    back = torch.zeros((NCHUNKS, CSZ), dtype=torch.uint8, device=dev)
    # ---- decompress: reference-written chunks, every chunk of the batch compared ----
    comp[:, :stock.size].copy_(torch.from_numpy(stock).to(dev).unsqueeze(0).expand(NCHUNKS, stock.size))
    bd = pkg.DeviceBatch([comp[i].data_ptr() for i in range(NCHUNKS)], [CSZ + 16] * NCHUNKS,
                         [back[i].data_ptr() for i in range(NCHUNKS)], [CSZ] * NCHUNKS)
    for rep in range(2):          # the second call runs with the first call's cost feedback
        back.zero_()
        assert bd.decompress() == 0
        assert bd.results() == [CSZ] * NCHUNKS
        assert torch.equal(back, src), (name, "decompress of reference-written chunks", rep)
    # ---- compress: the reference reads what the GPU wrote ----
    if gpu_enc is None:
        gpu_enc = lib.blosc_compname_to_compcode(codec.encode()) >= 0
    if not gpu_enc:
        return
    bc = pkg.DeviceBatch([src[i].data_ptr() for i in range(NCHUNKS)], [CSZ] * NCHUNKS,
                         [comp[i].data_ptr() for i in range(NCHUNKS)], [CSZ + 16] * NCHUNKS)
    for rep in range(2):
        comp.zero_()
        assert bc.compress(T, clevel, shuffle, codec.encode()) == 0
        cb = bc.results()
        assert all(c > 0 for c in cb)
        hs = header(comp[0][:16].cpu().numpy())
        assert hs == {**header(stock), "cbytes": cb[0]}, (hs, header(stock))     # same header policy as the reference
        for i in (0, NCHUNKS // 2, NCHUNKS - 1):
            ch = comp[i][:cb[i]].cpu().numpy()
            if ref is not None:
                rr, out = ref_decompress(ref, ch, CSZ)
            else:
                rr, out = orc_decompress(oracle, ch, CSZ)
            assert rr == CSZ and np.array_equal(out, data), (name, "reference cannot read GPU chunk", i, rep)
        # and our own decoder on the whole batch
        back.zero_()
        assert bd.decompress() == 0 and bd.results() == [CSZ] * NCHUNKS
        assert torch.equal(back, src)
