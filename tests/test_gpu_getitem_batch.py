"""blosc_getitem on the GPU and the device-resident batched API (include/blosc_gpu.h)."""
import numpy as np
import pytest

from helpers import DATASETS, orc_compress, orc_decompress, ptr

pytestmark = pytest.mark.gpu


def test_getitem_matches_oracle(pkg, lib, oracle):
    """tests/test_getitem.c + csv: every shuffle mode, slices inside / across blocks, whole buffer."""
    rng = np.random.default_rng(2)
    for codec in ["lz4", "blosclz"]:
        for shuffle in [0, 1, 2]:
            for T in [1, 4, 8, 17]:
                for n in [127, 1000, 300001, (1 << 21) + 5]:
                    n = n // T * T
                    if n == 0:
                        continue
                    data = DATASETS["bench19"](n)
                    _, chunk = orc_compress(oracle, data, T, 5, shuffle, codec)
                    ni = n // T
                    for (s, k) in [(0, ni), (0, 1), (ni - 1, 1), (ni // 3, ni // 2), (int(rng.integers(0, ni)), 0)]:
                        k = min(k, ni - s)
                        want = np.zeros(k * T + 1, np.uint8); got = np.zeros(k * T + 1, np.uint8)
                        r0 = oracle.orc_getitem(ptr(chunk), s, k, ptr(want))
                        r1 = lib.blosc_getitem(ptr(chunk), s, k, ptr(got))
                        assert r0 == r1 == k * T, (codec, shuffle, T, n, s, k, r0, r1)
                        assert np.array_equal(want, got)
    # out-of-range -> -1 (blosc.c:1645-1653)
    assert lib.blosc_getitem(ptr(chunk), -1, 1, ptr(got)) == -1
    assert lib.blosc_getitem(ptr(chunk), 0, ni + 1, ptr(got)) == -1


def test_device_batch_roundtrip(pkg, oracle):
    """Many chunks resident in HBM, mixed sizes/contents, one call each way; compressed chunks also
    decode on the CPU with the oracle; oracle-written chunks decode on the device."""
    import torch
    dev = torch.device("cuda:0")
    sizes = [1 << 20, 300001 * 8, 8 * 1000, 0, 100, (1 << 22) + 8, 1 << 16]
    names = ["bench19", "randwalk", "zeros", "random", "random", "linspace", "random"]
    host = [DATASETS[nm](n) for nm, n in zip(names, sizes)]
    src = [torch.from_numpy(h.copy()).to(dev) if h.size else torch.empty(0, dtype=torch.uint8, device=dev) for h in host]
    dst = [torch.empty(n + 16, dtype=torch.uint8, device=dev) for n in sizes]
    b = pkg.DeviceBatch([t.data_ptr() for t in src], sizes, [t.data_ptr() for t in dst], [n + 16 for n in sizes])
    assert b.compress(8, 5, 1, b"lz4") == 0
    cb = b.results()
    assert all(c > 0 for c in cb), cb
    chunks = [dst[i][:cb[i]].cpu().numpy() for i in range(len(sizes))]
    for h, c in zip(host, chunks):
        r, out = orc_decompress(oracle, c, h.size)
        assert r == h.size and np.array_equal(out, h)
    # device decompress of (a) the GPU chunks, (b) oracle chunks
    for variant in ("gpu", "oracle"):
        if variant == "oracle":
            chunks = [orc_compress(oracle, h, 8, 5, 1, "lz4")[1] for h in host]
        csrc = [torch.from_numpy(c.copy()).to(dev) for c in chunks]
        out = [torch.full((max(n, 1),), 0xEE, dtype=torch.uint8, device=dev) for n in sizes]
        b2 = pkg.DeviceBatch([t.data_ptr() for t in csrc], [c.size for c in chunks], [t.data_ptr() for t in out], sizes)
        assert b2.decompress() == 0
        assert b2.results() == sizes, (variant, b2.results())
        for h, o, n in zip(host, out, sizes):
            assert np.array_equal(o[:n].cpu().numpy(), h)
    # stock entry points accept device pointers too
    L = pkg.load()
    r = L.blosc_decompress_ctx(csrc[0].data_ptr(), out[0].data_ptr(), sizes[0], 1)
    assert r == sizes[0]
    # a bad chunk inside a batch only fails itself
    bad = chunks[1].copy(); bad[0] = 9
    csrc[1] = torch.from_numpy(bad).to(dev)
    b3 = pkg.DeviceBatch([t.data_ptr() for t in csrc], [c.size for c in chunks], [t.data_ptr() for t in out], sizes)
    assert b3.decompress() == 0
    res = b3.results()
    assert res[1] == -1 and res[0] == sizes[0] and res[2] == sizes[2]
