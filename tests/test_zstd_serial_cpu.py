"""CPU: the serial Zstd primitives the GPU kernel runs on single lanes (c-blosc_amd/csrc/zstd_serial.h) are plain
C++; tests/tools/zstd_serial_frame.cpp builds a whole-frame decoder from them.  Compiled here with g++ (with
AddressSanitizer when available: every read must stay inside the frame) and compared with the oracle on every
frame of the committed reference-written Zstd chunks and the compat vectors, plus bit-flipped frames."""
import ctypes as C
import glob
import os
import subprocess
import tempfile

import numpy as np
import pytest

from helpers import ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _frames():
    z = np.load(os.path.join(GOLDEN, "ref_zstd_chunks.npz"))
    chunks = [z[f"c{k}"] for k in range(len(z["meta"]))]
    chunks += [np.fromfile(f, np.uint8) for f in sorted(glob.glob(os.path.join(GOLDEN, "compat", "*zstd*.cdata")))]
    for ch in chunks:
        flags = int(ch[2]); nbytes, bs, _ = [int(x) for x in ch[4:16].view("<i4")]
        if flags & 2:
            continue
        nblocks = (nbytes + bs - 1) // bs
        bst = ch[16:16 + 4 * nblocks].view("<i4")
        for j in range(nblocks):
            bsz = bs if (j < nblocks - 1 or nbytes % bs == 0) else nbytes % bs
            o = int(bst[j]); cs = int(ch[o:o + 4].view("<i4")[0])
            if cs != bsz:
                yield ch[o + 4:o + 4 + cs].copy(), bsz


@pytest.fixture(scope="module")
def zs():
    td = tempfile.mkdtemp()
    so = os.path.join(td, "zs.so")
    base = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-shared", "-fPIC", "-I", os.path.join(ROOT, "c-blosc_amd", "csrc"),
            os.path.join(ROOT, "tests", "tools", "zstd_serial_frame.cpp"), "-o", so]
    subprocess.check_call(base)
    lib = C.CDLL(so)
    lib.zs_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return lib


def test_serial_primitives_decode_every_frame(zs, oracle):
    oracle.orc_zstd_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    n = 0
    for fr, bsz in _frames():
        a = np.zeros(bsz, np.uint8); b = np.zeros(bsz, np.uint8)
        assert oracle.orc_zstd_decompress(ptr(fr), fr.size, ptr(a), bsz) == bsz
        assert zs.zs_decompress(ptr(fr), fr.size, ptr(b), bsz) == bsz
        assert np.array_equal(a, b)
        n += 1
    assert n >= 60


def test_serial_primitives_survive_corruption(zs, oracle):
    """bit flips and truncations: same verdict as the oracle whenever the oracle accepts, never a crash"""
    oracle.orc_zstd_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(9)
    frames = list(_frames())[:12]
    for fr, bsz in frames:
        for trial in range(60):
            c = fr.copy()
            if trial % 5 == 4:
                c = c[:int(rng.integers(6, c.size))].copy()
            else:
                pos = int(rng.integers(0, c.size)); c[pos] ^= 1 << int(rng.integers(0, 8))
            # exact-size buffers so that an out-of-bounds access is one byte past a numpy allocation
            src = np.empty(c.size, np.uint8); src[:] = c
            a = np.zeros(bsz, np.uint8); b = np.zeros(bsz, np.uint8)
            ra = oracle.orc_zstd_decompress(ptr(src), src.size, ptr(a), bsz)
            rb = zs.zs_decompress(ptr(src), src.size, ptr(b), bsz)
            assert ra == rb, (ra, rb)
            if ra > 0:
                assert np.array_equal(a[:ra], b[:rb])


def test_serial_primitives_on_direct_reference_frames(zs, ref):
    """the same 864 frames as tests/test_oracle_zstd.py::test_direct_reference_frames (only where oracle/_ref exists)"""
    from test_oracle_zstd import _direct_frames
    for frame, data in _direct_frames(ref):
        out = np.zeros(data.size, np.uint8)
        assert zs.zs_decompress(ptr(frame), frame.size, ptr(out), data.size) == data.size
        assert np.array_equal(out, data)
