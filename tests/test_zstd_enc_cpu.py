"""CPU: the format-writing half of the GPU Zstd encoder (c-blosc_amd/csrc/zstd_enc.h: frame / block / raw-literals
headers, sequence codes, predefined FSE tables as the encoder sees them, repeat-offset codes, the backward bitstream)
compiled with g++ behind a plain greedy matcher (tests/tools/zstd_enc_cpu.cpp).  Every frame must be read back
bit-exactly by the oracle's decoder and - where oracle/_ref ships - by the reference's own ZSTD_decompress."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import DATASETS, ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def enc():
    so = os.path.join(ROOT, "tests", "tools", "libzstd_enc_cpu.so")
    src = os.path.join(ROOT, "tests", "tools", "zstd_enc_cpu.cpp")
    hdr = os.path.join(ROOT, "c-blosc_amd", "csrc", "zstd_enc.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, src])
    E = C.CDLL(so)
    E.zenc_cpu_compress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    return E


def test_frames_decode_with_oracle_and_reference(enc, oracle, ref):
    oracle.orc_zstd_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    if ref is not None:
        ref.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        ref.ZSTD_decompress.restype = C.c_size_t
        ref.ZSTD_isError.argtypes = [C.c_size_t]
    cases = 0
    for dname in ["bench19", "linspace", "randwalk", "smallints", "zeros", "random"]:
        for n in [1, 5, 100, 1000, 4096, 65536, 131072, 131073, 300001, 1 << 20]:      # 131073+: several blocks per frame
            data = DATASETS[dname](n)
            if dname != "random" and n >= 4096 and n % 8 == 0:
                data = data.reshape(-1, 8).T.copy().reshape(-1)                          # byte planes, as inside a blosc block
            for minmatch in (3, 4, 8):
                out = np.zeros(n + n // 8 + 64, np.uint8)
                r = enc.zenc_cpu_compress(ptr(data), n, ptr(out), out.size, minmatch)
                assert r > 0
                back = np.zeros(n + 8, np.uint8)
                assert oracle.orc_zstd_decompress(ptr(out), r, ptr(back), n) == n and np.array_equal(back[:n], data), (dname, n, minmatch)
                if ref is not None:
                    back2 = np.zeros(n + 8, np.uint8)
                    d = ref.ZSTD_decompress(ptr(back2), n, ptr(out), r)
                    assert not ref.ZSTD_isError(d) and d == n and np.array_equal(back2[:n], data), (dname, n, minmatch)
                cases += 1
    assert cases == 180


def test_too_small_destination(enc):
    data = DATASETS["random"](5000)
    out = np.zeros(8000, np.uint8)
    assert enc.zenc_cpu_compress(ptr(data), 5000, ptr(out), 4000, 4) == 0        # does not fit: the caller stores the split raw
