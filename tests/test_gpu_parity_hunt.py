"""GPU: a seeded slice of scripts/parity_hunt.py inside the suite - BLOSC_HUNT_CELLS (default 300) random cells of
19 typesizes x 3 filters x 5 codecs x 9 compression levels x 8 data sets, every batch made of DIFFERENT chunks (other offsets into the data set,
other - also odd - sizes, so chunks end in the middle of an element and of a block and the queues deal planes of different cost side by side):
  * chunks written by the reference, decoded here, every byte compared (twice: the second call runs in the cost-feedback order);
  * the same inputs compressed here, read by the reference (two chunks of the batch) and by this library (all of them).
BLOSC_HUNT_SEED (default 0) draws another slice; the failing cells are listed with everything needed to run them again.  Round 5's silent
decode error (right sizes, wrong bytes) sat in one cell of this grid and was found by an A/B script, not by the suite: this test is the suite's
own net for that class.  The grid of tests/test_compress_roundtrip.csv (the reference's own) is what it samples, widened by typesize and data."""
import os

import numpy as np
import pytest

from helpers import DATASETS, ref_compress, ref_decompress

pytestmark = pytest.mark.gpu
TYPESIZES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 64, 255]
CODECS = ["lz4", "lz4", "lz4", "blosclz", "blosclz", "lz4hc", "zstd", "zlib"]      # (drawn uniformly from this list: the LZ codecs most often)
NCH = 6
POOL = 24 << 20            # bytes of every data set the chunks are cut from
MAXN = 6 << 20


def test_seeded_slice_of_the_parity_grid(pkg, lib, ref):
    import torch
    if ref is None:
        pytest.skip("needs the reference (oracle/_ref) as writer and reader")
    seed = int(os.environ.get("BLOSC_HUNT_SEED", "0"))
    cells = int(os.environ.get("BLOSC_HUNT_CELLS", "300"))
    rng = np.random.default_rng(1000003 * seed + 17)
    dev = torch.device("cuda:0")
    pools = {d: DATASETS[d](POOL) for d in DATASETS}
    d_pools = {d: torch.from_numpy(p).to(dev) for d, p in pools.items()}
    comp = torch.zeros((NCH, MAXN + 256), dtype=torch.uint8, device=dev)
    back = torch.zeros((NCH, MAXN), dtype=torch.uint8, device=dev)
    bad = []
    for cell in range(cells):
        dname = str(rng.choice(list(DATASETS)))
        T = int(rng.choice(TYPESIZES))
        shuffle = int(rng.choice([1, 1, 2, 2, 0]))
        codec = str(rng.choice(CODECS))
        clevel = int(rng.integers(1, 10))
        # sizes: whole MiB, odd tails, small chunks (a single block), one chunk of the batch's maximum
        sizes = []
        for i in range(NCH):
            kind = int(rng.integers(0, 4))
            n = [int(rng.integers(1, 7)) << 20, int(rng.integers(1 << 20, MAXN)), int(rng.integers(1, 200000)), MAXN][kind]
            sizes.append(max(1, min(n, MAXN)))
        offs = [int(rng.integers(0, (POOL - n) // 8 + 1)) * 8 for n in sizes]
        what = dict(seed=seed, cell=cell, data=dname, T=T, shuffle=shuffle, codec=codec, clevel=clevel, sizes=sizes, offs=offs)
        datas = [pools[dname][o:o + n] for o, n in zip(offs, sizes)]
        d_srcs = [d_pools[dname][o:o + n] for o, n in zip(offs, sizes)]
        # ---- reference-written chunks -> here ----
        stocks = []
        for d in datas:
            r, st = ref_compress(ref, d, T, clevel, shuffle, codec.encode(), nthreads=4)
            assert r > 0, what
            stocks.append(st)
        for i, st in enumerate(stocks):
            comp[i, :st.size].copy_(torch.from_numpy(st).to(dev))
        bd = pkg.DeviceBatch([comp[i].data_ptr() for i in range(NCH)], [st.size for st in stocks], [back[i].data_ptr() for i in range(NCH)], sizes)
        for rep in range(2):
            back.fill_(0xEE)
            rc = bd.decompress()
            res = bd.results()
            wrong = [i for i in range(NCH) if res[i] != sizes[i] or not bool(torch.equal(back[i, :sizes[i]], d_srcs[i]))]
            spill = [i for i in range(NCH) if sizes[i] < MAXN and int(back[i, sizes[i]]) != 0xEE]
            if rc != 0 or wrong or spill:
                bad.append(("decode of reference-written chunks", what, rep, rc, res, wrong, spill))
        # ---- here -> the reference and this library ----
        bc = pkg.DeviceBatch([d.data_ptr() for d in d_srcs], sizes, [comp[i].data_ptr() for i in range(NCH)], [n + 16 for n in sizes])
        rc = bc.compress(T, clevel, shuffle, codec.encode())
        cb = bc.results()
        if rc != 0 or min(cb) <= 0 or any(c > n + 16 for c, n in zip(cb, sizes)):
            bad.append(("compress", what, rc, cb))
            continue
        for i in sorted({int(rng.integers(0, NCH)), NCH - 1}):
            rr, out = ref_decompress(ref, comp[i][:cb[i]].cpu().numpy(), sizes[i])
            if rr != sizes[i] or not np.array_equal(out[:sizes[i]], datas[i]):
                bad.append(("the reference reading a chunk written here", what, i, rr))
        bd2 = pkg.DeviceBatch([comp[i].data_ptr() for i in range(NCH)], cb, [back[i].data_ptr() for i in range(NCH)], sizes)
        back.fill_(0xEE)
        rc = bd2.decompress()
        res = bd2.results()
        wrong = [i for i in range(NCH) if res[i] != sizes[i] or not bool(torch.equal(back[i, :sizes[i]], d_srcs[i]))]
        if rc != 0 or wrong:
            bad.append(("decode of own chunks", what, rc, res, wrong))
    assert not bad, (len(bad), bad[:6])
