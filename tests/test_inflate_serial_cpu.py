"""CPU: the serial zlib primitives the GPU kernel runs (c-blosc_amd/csrc/inflate_serial.h) are plain C++;
tests/tools/inflate_serial_stream.cpp builds a whole-stream decoder from them.  Compiled here with g++ and compared with the
oracle on stock, hand-built and corrupted streams, and - where oracle/_ref exists - with the reference's own `uncompress`."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import deflate_builder as D
from test_oracle_zlib import HANDBUILT, _un, _zo, mutate, ref_uncompress, stock_streams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def zi():
    so = os.path.join(tempfile.mkdtemp(), "zi.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-shared", "-fPIC", "-I", os.path.join(ROOT, "c-blosc_amd", "csrc"),
                           os.path.join(ROOT, "tests", "tools", "inflate_serial_stream.cpp"), "-o", so])
    lib = C.CDLL(so)
    lib.zi_uncompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return lib.zi_uncompress


def test_serial_primitives_on_stock_streams(zi):
    n = 0
    for s, d in stock_streams():
        r, out = _un(zi, s, d.size)
        assert r == d.size and np.array_equal(out[:r], d)
        n += 1
    assert n >= 300


def test_serial_primitives_on_handbuilt_streams(zi):
    for name, s, cap in D.cases():
        assert _un(zi, s, max(cap, 1) + 300)[0] == HANDBUILT[name], name


def test_serial_primitives_survive_corruption(zi, oracle, ref):
    """same verdict and bytes as the oracle (and as the reference, where it is built) on truncations, extensions, bit flips"""
    zo = _zo(oracle); ru = ref_uncompress(ref) if ref is not None else None
    rng = np.random.default_rng(17); ntot = 0
    for s, d in stock_streams(sizes=(3, 255, 4096, 70000)):
        for t in range(12 if s.size < 20000 else 4):
            c = mutate(s, t, rng)
            r, out = _un(zi, c, d.size); ro, oo = _un(zo, c, d.size)
            assert r == ro and np.array_equal(out[:r], oo[:ro]), (t, s.size, r, ro)
            if ru is not None:
                rr, orr = _un(ru, c, d.size)
                assert r == rr and np.array_equal(out[:r], orr[:rr]), (t, s.size, r, rr)
            ntot += 1
    assert ntot > 1500
