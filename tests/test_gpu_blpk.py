"""GPU: whole files through the bloscpack-layout container (c-blosc_amd/blpk.py): chunks of a file go through the
BATCHED calls with host buffers; every chunk inside a written file is an ordinary c-blosc chunk (read here by the oracle and the
reference), and a file assembled from chunks the REFERENCE wrote is unpacked bit-exactly."""
import importlib.util
import io
import os
import struct

import numpy as np
import pytest

from helpers import DATASETS, orc_decompress, ref_compress, ref_decompress

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def blpk():
    spec = importlib.util.spec_from_file_location("blpk", os.path.join(ROOT, "c-blosc_amd", "blpk.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("cname", [b"lz4", b"blosclz", b"zstd", b"zlib"])
def test_file_roundtrip_and_chunks_are_stock(blpk, lib, oracle, ref, cname, tmp_path):
    for dname, n, cs, T, checksum in (("bench19", (5 << 20) + 12345, 1 << 20, 8, 1), ("randwalk", 3 << 20, 1 << 19, 8, 2), ("zeros", 100, 1 << 20, 1, 0),
                                      ("smallints", (1 << 22) + 4, 700001, 4, 1)):
        data = DATASETS[dname](n)
        path = tmp_path / f"{dname}.blp"
        with open(path, "wb") as fh:
            nchunks, nbytes = blpk.pack(lib, data, fh, chunk_size=cs, typesize=T, clevel=5, shuffle=1, cname=cname, checksum=checksum, batch_bytes=2 << 20)
        assert nchunks == (n + cs - 1) // cs and os.path.getsize(path) == nbytes
        with open(path, "rb") as fh:
            back = blpk.unpack(lib, fh, batch_bytes=2 << 20)
        assert np.array_equal(back, data), (dname, cname)
        # the chunks inside are plain c-blosc chunks: walk the offset table, decode each with the checkers
        blob = open(path, "rb").read()
        h = blpk.unpack_header(blob)
        offs = np.frombuffer(blob, "<i8", nchunks, 32)
        for k in range(nchunks):
            o = int(offs[k]); nb, _, cb = struct.unpack("<iii", blob[o + 4:o + 16])
            chunk = np.frombuffer(blob, np.uint8, cb, o).copy()
            want = data[k * cs:min(n, (k + 1) * cs)]
            r, out = orc_decompress(oracle, chunk, nb)
            assert r == want.size and np.array_equal(out, want), (dname, cname, k)
            if ref is not None:
                r2, out2 = ref_decompress(ref, chunk, nb)
                assert r2 == want.size and np.array_equal(out2, want)


def test_corrupt_file_is_refused(blpk, lib, tmp_path):
    data = DATASETS["bench19"](3 << 20)
    buf = io.BytesIO()
    blpk.pack(lib, data, buf, chunk_size=1 << 20, typesize=8, cname=b"lz4", checksum=1)
    blob = bytearray(buf.getvalue())
    blob[-100] ^= 0x40                                   # inside the last chunk: its adler32 no longer fits
    with pytest.raises(blpk.BlpkError):
        blpk.unpack(lib, io.BytesIO(bytes(blob)))


def test_file_of_reference_written_chunks(blpk, lib, ref):
    """a bloscpack-layout file whose chunks were written by the REFERENCE (what a stock bloscpack would hold)"""
    if ref is None:
        pytest.skip("needs oracle/_ref")
    import zlib as _z
    data = DATASETS["linspace"]((4 << 20) + 999)
    cs = 1 << 20
    chunks = []
    for k in range(0, data.size, cs):
        r, c = ref_compress(ref, data[k:k + cs], 8, 7, 1, b"blosclz")
        assert r > 0
        chunks.append(c[:r].tobytes())
    n = len(chunks)
    hdr = blpk.pack_header(n, cs, data.size - (n - 1) * cs, 8, checksum=1)
    pos = 32 + 8 * n; offs = []; body = b""
    for c in chunks:
        offs.append(pos + len(body)); body += c + struct.pack("<I", _z.adler32(c) & 0xffffffff)
    blob = hdr + np.array(offs, "<i8").tobytes() + body
    back = blpk.unpack(lib, io.BytesIO(blob))
    assert np.array_equal(back, data)
