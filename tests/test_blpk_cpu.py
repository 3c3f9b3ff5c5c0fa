"""CPU: the bloscpack-layout container (c-blosc_amd/blpk.py) - header and offset-table handling without a GPU."""
import importlib.util
import io
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def blpk():
    spec = importlib.util.spec_from_file_location("blpk", os.path.join(ROOT, "c-blosc_amd", "blpk.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_header_layout(blpk):
    h = blpk.pack_header(nchunks=5, chunk_size=1 << 20, last_chunk=12345, typesize=8, checksum=1)
    assert len(h) == 32 and h[:4] == b"blpk" and h[4] == 3 and h[5] == 1 and h[6] == 1 and h[7] == 8
    assert struct.unpack("<i", h[8:12])[0] == 1 << 20 and struct.unpack("<i", h[12:16])[0] == 12345
    assert struct.unpack("<q", h[16:24])[0] == 5 and struct.unpack("<q", h[24:32])[0] == 0
    d = blpk.unpack_header(h)
    assert d["nchunks"] == 5 and d["chunk_size"] == 1 << 20 and d["last_chunk"] == 12345 and d["typesize"] == 8 and d["offsets"] and not d["metadata"]


def test_bad_headers(blpk):
    good = blpk.pack_header(1, 100, 100, 1)
    with pytest.raises(blpk.BlpkError):
        blpk.unpack_header(good[:20])
    with pytest.raises(blpk.BlpkError):
        blpk.unpack_header(b"blpX" + good[4:])
    with pytest.raises(blpk.BlpkError):
        blpk.unpack_header(good[:4] + bytes([2]) + good[5:])
    with pytest.raises(blpk.BlpkError):
        blpk.unpack_header(good[:6] + bytes([7]) + good[7:])


def test_reader_rejects_offsets_outside_the_file(blpk):
    h = blpk.pack_header(1, 100, 100, 1, checksum=0)
    blob = h + struct.pack("<q", 10**9)
    with pytest.raises(blpk.BlpkError):
        blpk.unpack(None, io.BytesIO(blob))
    blob = h + struct.pack("<q", 40) + bytes(8)          # chunk header cut short
    with pytest.raises(blpk.BlpkError):
        blpk.unpack(None, io.BytesIO(blob))
