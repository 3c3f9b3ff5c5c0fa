"""GPU: Zstd frames with sequence tables made per block (BLOSC_AMD_ZSTD_TABLES=1, k_encode.hip: zt_make_tables)
instead of the predefined ones.  Same contract as every encoder here: stock c-blosc (ZSTD_decompress), the oracle and our own
decoder read every chunk back bit-exactly; and the point of it - the ratio - is checked against the predefined-table frames.
The CPU suite runs the same device source through the wavefront emulator (tests/test_wave_emu_encoders.py)."""
import os

import numpy as np
import pytest

from helpers import DATASETS, header, orc_decompress, ref_compress, ref_decompress

pytestmark = pytest.mark.gpu


@pytest.fixture()
def tables_on():
    old = os.environ.get("BLOSC_AMD_ZSTD_TABLES")
    os.environ["BLOSC_AMD_ZSTD_TABLES"] = "1"
    yield
    if old is None:
        del os.environ["BLOSC_AMD_ZSTD_TABLES"]
    else:
        os.environ["BLOSC_AMD_ZSTD_TABLES"] = old


def _roundtrip(pkg, oracle, ref, data, T, clevel, shuffle, blocksize=0):
    r, chunk = pkg.compress(data, T, clevel, shuffle, b"zstd", blocksize)
    assert 0 < r <= data.size + 16
    assert header(chunk)["cbytes"] == r
    r2, out = orc_decompress(oracle, chunk, data.size)
    assert r2 == data.size and np.array_equal(out, data)
    if ref is not None:
        r3, out3 = ref_decompress(ref, chunk, data.size)
        assert r3 == data.size and np.array_equal(out3, data)
    r4, out4 = pkg.decompress(chunk, data.size)
    assert r4 == data.size and np.array_equal(out4, data)
    return r


def test_frames_with_block_tables_are_valid_everywhere(pkg, oracle, ref, tables_on):
    for dname in ["bench19", "linspace", "randwalk", "smallints", "zeros", "random", "arange"]:
        for T, shuffle in [(8, 1), (4, 1), (4, 2), (1, 0), (3, 1)]:
            for n in [129, 1000, 32768, 65536 + 17, 300001, (1 << 21) + 5]:
                if n > 400000 and (T != 8 or dname in ("zeros", "random", "arange")):
                    continue
                for clevel in ((1, 3, 9) if n <= 32768 else (3,)):
                    _roundtrip(pkg, oracle, ref, DATASETS[dname](n), T, clevel, shuffle)
    for bs in (128, 512, 4096, 1 << 20):            # tiny blocks (RLE tables, nothing to code) and frames of several Zstd blocks
        _roundtrip(pkg, oracle, ref, DATASETS["bench19"](300001), 1, 3, 0, blocksize=bs)
        _roundtrip(pkg, oracle, ref, DATASETS["linspace"](300001 * 8), 8, 3, 1, blocksize=bs * 8)


def test_block_tables_ratio(pkg, oracle, ref, tables_on):
    for dname, T, want in [("bench19", 8, 0.85), ("linspace", 8, 0.85), ("randwalk", 8, 1.005), ("smallints", 4, 1.005)]:
        data = DATASETS[dname](16 << 20)
        r = _roundtrip(pkg, oracle, ref, data, T, 3, 1)
        os.environ["BLOSC_AMD_ZSTD_TABLES"] = "0"
        rp, _ = pkg.compress(data, T, 3, 1, b"zstd", 0)
        os.environ["BLOSC_AMD_ZSTD_TABLES"] = "1"
        rr = ref_compress(ref, data, T, 3, 1, b"zstd", nthreads=8)[0] if ref is not None else 0
        print(f"zstd clevel 3 {dname:9s} T={T}: per-block tables {data.size / r:8.2f}   predefined tables {data.size / rp:8.2f}   reference {data.size / rr if rr else 0:8.2f}")
        assert r <= rp * want, (dname, r, rp)


@pytest.mark.parametrize("codec,switch", [(b"zstd", "BLOSC_AMD_ZSTD_SEARCH"), (b"zlib", "BLOSC_AMD_ZLIB_SEARCH")], ids=["zstd", "zlib"])
def test_lz4hc_search_in_front_of_the_entropy_writers(pkg, oracle, ref, codec, switch):
    """BLOSC_AMD_ZSTD_SEARCH=1 / BLOSC_AMD_ZLIB_SEARCH=1: the LZ4HC-grade search of DESIGN.md 3.9 feeds the Zstd writer (with per-block
    tables) / the zlib writer.  Chunks must be read by everybody; the ratio is printed next to the plain path's and the reference's."""
    old = os.environ.get(switch)
    try:
        for dname, T in [("bench19", 8), ("linspace", 8), ("smallints", 4), ("randwalk", 8), ("zeros", 8)]:
            for n in (1000, 65536 + 17, 300001, 4 << 20):
                data = DATASETS[dname](n)
                os.environ[switch] = "1"
                r, chunk = pkg.compress(data, T, 5, 1, codec, 0)
                os.environ[switch] = "0"
                rp, _ = pkg.compress(data, T, 5, 1, codec, 0)
                assert 0 < r <= data.size + 16 and header(chunk)["cbytes"] == r
                r2, out = orc_decompress(oracle, chunk, data.size)
                assert r2 == data.size and np.array_equal(out, data)
                if ref is not None:
                    r3, out3 = ref_decompress(ref, chunk, data.size)
                    assert r3 == data.size and np.array_equal(out3, data)
                r4, out4 = pkg.decompress(chunk, data.size)
                assert r4 == data.size and np.array_equal(out4, data)
                if n == 4 << 20:
                    rr = ref_compress(ref, data, T, 5, 1, codec, nthreads=8)[0] if ref is not None else 0
                    print(f"{codec.decode()} clevel 5 {dname:9s}: with the search {data.size / r:8.2f}   plain {data.size / rp:8.2f}   reference {data.size / rr if rr else 0:8.2f}")
                    assert r <= rp * 1.06            # (all-zero data: a byte or two of table modes per block on top of almost nothing)
    finally:
        if old is None:
            os.environ.pop(switch, None)
        else:
            os.environ[switch] = old


def test_huffman_literals(pkg, oracle, ref):
    """BLOSC_AMD_ZSTD_HUFFMAN=1 on top of the per-block tables (and of the search): literal-heavy data must come out smaller and be read by
    everybody (the reference's ZSTD_decompress checks the Huffman tree description and the four streams)."""
    keys = ("BLOSC_AMD_ZSTD_TABLES", "BLOSC_AMD_ZSTD_SEARCH", "BLOSC_AMD_ZSTD_HUFFMAN")
    old = {k: os.environ.get(k) for k in keys}
    try:
        for base in ("BLOSC_AMD_ZSTD_TABLES", "BLOSC_AMD_ZSTD_SEARCH"):
            for k in keys:
                os.environ.pop(k, None)
            os.environ[base] = "1"
            for dname, T, want in [("smallints", 4, 0.92), ("randwalk", 8, 1.002), ("bench19", 8, 1.002), ("random", 1, 1.002)]:
                for n in (1000, 300001, 4 << 20):
                    data = DATASETS[dname](n)
                    os.environ["BLOSC_AMD_ZSTD_HUFFMAN"] = "0"
                    rp, _ = pkg.compress(data, T, 3, 1, b"zstd", 0)
                    os.environ["BLOSC_AMD_ZSTD_HUFFMAN"] = "1"
                    r = _roundtrip(pkg, oracle, ref, data, T, 3, 1)
                    if n == 4 << 20:
                        print(f"zstd clevel 3 {dname:9s} ({base[-6:].lower()}): Huffman literals {data.size / r:8.2f}   raw literals {data.size / rp:8.2f}")
                        assert r <= rp * want, (dname, base, r, rp)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_zlib_dynamic_huffman_codes(pkg, oracle, ref):
    """BLOSC_AMD_ZLIB_DYNAMIC=1 (alone and with BLOSC_AMD_ZLIB_SEARCH=1): one final block with Huffman codes made for the stream
    (DESIGN.md 3.8).  Stock zlib (the reference's uncompress), the oracle and our own decoder read every chunk."""
    keys = ("BLOSC_AMD_ZLIB_DYNAMIC", "BLOSC_AMD_ZLIB_SEARCH")
    old = {k: os.environ.get(k) for k in keys}
    try:
        for search in ("0", "1"):
            for dname, T in [("bench19", 8), ("linspace", 8), ("smallints", 4), ("randwalk", 8), ("zeros", 8), ("random", 1)]:
                for n in (129, 1000, 65536 + 17, 300001, 4 << 20):
                    data = DATASETS[dname](n)
                    os.environ["BLOSC_AMD_ZLIB_SEARCH"] = search
                    os.environ["BLOSC_AMD_ZLIB_DYNAMIC"] = "0"
                    rp, _ = pkg.compress(data, T, 5, 1, b"zlib", 0)
                    os.environ["BLOSC_AMD_ZLIB_DYNAMIC"] = "1"
                    r, chunk = pkg.compress(data, T, 5, 1, b"zlib", 0)
                    assert 0 < r <= data.size + 16 and header(chunk)["cbytes"] == r
                    r2, out = orc_decompress(oracle, chunk, data.size)
                    assert r2 == data.size and np.array_equal(out, data)
                    if ref is not None:
                        r3, out3 = ref_decompress(ref, chunk, data.size)
                        assert r3 == data.size and np.array_equal(out3, data)
                    r4, out4 = pkg.decompress(chunk, data.size)
                    assert r4 == data.size and np.array_equal(out4, data)
                    if n == 4 << 20:
                        rr = ref_compress(ref, data, T, 5, 1, b"zlib", nthreads=8)[0] if ref is not None else 0
                        print(f"zlib clevel 5 {dname:9s} search={search}: dynamic codes {data.size / r:8.2f}   fixed codes {data.size / rp:8.2f}   reference {data.size / rr if rr else 0:8.2f}")
                        assert r <= rp * 1.002
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
