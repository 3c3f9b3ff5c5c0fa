"""CPU: the product library loads, exports every symbol include/*.h declares, and its host-side logic
(names, codes, introspection, policy) matches the reference.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu
from helpers import ptr


def _declared(path):
    txt = open(path).read()
    return sorted(set(re.findall(r"BLOSC_EXPORT[^;(]*?\b(blosc_\w+)\s*\(", txt)))


def test_exports_every_declared_symbol(lib, pkg):
    names = _declared(os.path.join(ROOT, "include", "blosc.h")) + _declared(os.path.join(ROOT, "include", "blosc_gpu.h"))
    assert len(names) == 25 + 13
    assert sorted(names) == sorted(pkg.STOCK_SYMBOLS + pkg.GPU_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_names_codes_versions(lib):
    assert lib.blosc_get_version_string() == b"1.21.7.dev"
    assert lib.blosc_list_compressors() == b"blosclz,lz4,lz4hc,zlib,zstd"
    for name, code in [(b"blosclz", 0), (b"lz4", 1), (b"lz4hc", 2), (b"zlib", 4), (b"zstd", 5)]:
        assert lib.blosc_compname_to_compcode(name) == code
        p = C.c_char_p()
        assert lib.blosc_compcode_to_compname(code, C.byref(p)) == code and p.value == name
    for name in [b"snappy", b"nope"]:
        assert lib.blosc_compname_to_compcode(name) == -1
    p = C.c_char_p()
    assert lib.blosc_compcode_to_compname(3, C.byref(p)) == -1 and p.value == b"snappy"   # name known, support absent
    a, b = C.c_char_p(), C.c_char_p()
    assert lib.blosc_get_complib_info(b"lz4", C.byref(a), C.byref(b)) == 1 and a.value == b"LZ4"
    assert lib.blosc_get_complib_info(b"zstd", C.byref(a), C.byref(b)) == 4 and a.value == b"Zstd"
    assert lib.blosc_get_complib_info(b"zlib", C.byref(a), C.byref(b)) == 3 and a.value == b"Zlib"
    assert lib.blosc_get_complib_info(b"snappy", C.byref(a), C.byref(b)) == -1


def test_globals(lib):
    lib.blosc_init()
    assert lib.blosc_set_nthreads(4) == 1 and lib.blosc_get_nthreads() == 4 and lib.blosc_set_nthreads(1) == 4
    assert lib.blosc_get_compressor() == b"blosclz"
    assert lib.blosc_set_compressor(b"lz4") == 1 and lib.blosc_get_compressor() == b"lz4"
    assert lib.blosc_set_compressor(b"blosclz") == 0
    lib.blosc_set_blocksize(4096); assert lib.blosc_get_blocksize() == 4096; lib.blosc_set_blocksize(0)
    assert lib.blosc_free_resources() == 0
    lib.blosc_destroy()
    assert lib.blosc_free_resources() == -1      # not initialised (blosc.c:2313-2314)


def test_cbuffer_introspection_on_golden(lib):
    chunk = np.fromfile(os.path.join(ROOT, "tests", "golden", "compat", "blosc-1.18.0-lz4-bitshuffle.cdata"), np.uint8)
    nb, cb, bs = C.c_size_t(), C.c_size_t(), C.c_size_t()
    lib.blosc_cbuffer_sizes(ptr(chunk), C.byref(nb), C.byref(cb), C.byref(bs))
    assert (nb.value, cb.value, bs.value) == (4000000, chunk.size, 262144)
    assert lib.blosc_cbuffer_validate(ptr(chunk), chunk.size, C.byref(nb)) == 0
    assert lib.blosc_cbuffer_validate(ptr(chunk), chunk.size - 1, C.byref(nb)) == -1
    ts, fl = C.c_size_t(), C.c_int()
    lib.blosc_cbuffer_metainfo(ptr(chunk), C.byref(ts), C.byref(fl))
    assert ts.value == 4 and fl.value == 4          # BLOSC_DOBITSHUFFLE
    v, vl = C.c_int(), C.c_int()
    lib.blosc_cbuffer_versions(ptr(chunk), C.byref(v), C.byref(vl))
    assert (v.value, vl.value) == (2, 1)
    assert lib.blosc_cbuffer_complib(ptr(chunk)) == b"LZ4"


def test_policy_equals_oracle(lib, oracle):
    """compute_blocksize / split_block (blosc.c:929-1060) for every codec, clevel, split mode."""
    bad = 0
    for codec in range(6):
        for clevel in range(10):
            for T in [1, 2, 3, 4, 7, 8, 16, 17, 32, 255]:
                for n in [0, 1, 7, 100, 128, 1000, 32767, 32768, 100000, 1 << 20, (1 << 26) + 3, 2**31 - 17]:
                    for forced in [0, 1, 100, 4096 + T, 1 << 20, 2**31 - 1]:
                        for sm in [1, 2, 3, 4]:
                            a = lib.blosc_amd_policy_blocksize(clevel, T, n, forced, codec, sm)
                            b = oracle.orc_compute_blocksize(clevel, T, n, forced, codec, sm)
                            bad += a != b
                            if a > 0:
                                bad += lib.blosc_amd_policy_split(codec, T, a, sm) != oracle.orc_split_block(codec, T, a, sm)
    assert bad == 0


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_gpu_fails_loudly(lib):
    """No CPU fallback: without a device every compute entry point reports an error."""
    data = np.arange(1000, dtype="<i4").view(np.uint8)
    out = np.zeros(data.size + 16, np.uint8)
    assert lib.blosc_compress_ctx(5, 1, 4, data.size, ptr(data), ptr(out), out.size, b"lz4", 0, 1) == -1
    chunk = np.fromfile(os.path.join(ROOT, "tests", "golden", "compat", "blosc-1.18.0-lz4.cdata"), np.uint8)
    assert lib.blosc_decompress_ctx(ptr(chunk), ptr(np.zeros(4000000, np.uint8)), 4000000, 1) == -1
    assert lib.blosc_getitem(ptr(chunk), 0, 10, ptr(out)) == -1


def test_rccl_exchange_library_exports_its_header():
    """include/blosc_gpu_rccl.h: every declared entry point is exported by c-blosc_amd/libblosc_amd_rccl.so (a library of its own: the drop-in must
    not depend on RCCL), and the drop-in does not link RCCL."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "blosc_gpu_rccl.h")).read()
    names = set(re.findall(r"BLOSC_EXPORT\s+\w[\w\s\*]*?\b(blosc_gpu_\w+)\s*\(", hdr))
    assert len(names) == 10, names
    so = os.path.join(root, "c-blosc_amd", "libblosc_amd_rccl.so")
    assert os.path.exists(so), "build it: make -C c-blosc_amd"
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    for n in names:
        assert re.search(rf"\bT {n}\b", syms), n
    needed = subprocess.run(["readelf", "-d", os.path.join(root, "c-blosc_amd", "libblosc_amd.so")], capture_output=True, text=True, check=True).stdout
    assert "rccl" not in needed
