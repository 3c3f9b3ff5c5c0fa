"""Round trips under whatever BLOSC_AMD_* switches the environment carries (tests/test_gpu_modes.py runs this
script once per combination: the switches are read once per process).  No torch: host buffers through the stock
ABI.  Prints 'modes ok <n>' or raises.  BLOSC_MODE_CHECK_SHRINK=k divides the input sizes by k (the emulated library of
tests/tools/blosc_emu_lib.cpp runs the same combinations on the CPU, tests/test_emu_library.py)."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS, orc_compress, orc_decompress, ptr, wrap_planes_as_chunk  # noqa: E402

spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
pkg.load()
O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
sz, i, vp = C.c_size_t, C.c_int, C.c_void_p
O.orc_compress.argtypes = [i, i, sz, sz, vp, vp, sz, i, sz, i]
O.orc_decompress.argtypes = [vp, vp, sz]
O.orc_lz4_compress.argtypes = [vp, i, vp, i, i]

SHRINK = int(os.environ.get("BLOSC_MODE_CHECK_SHRINK", "1"))
n_ok = 0
for codec in ("lz4", "blosclz"):
    for dname, T, shuffle, n in [("bench19", 8, 1, (8 << 20) + 40), ("bench19", 4, 1, 3 << 20), ("linspace", 8, 1, 4 << 20),
                                 ("randwalk", 8, 1, 2 << 20), ("bench19", 4, 2, 2 << 20), ("zeros", 8, 1, 4 << 20),
                                 ("bench19", 2, 1, 1 << 20), ("bench19", 16, 1, 2 << 20), ("smallints", 4, 0, 1 << 20)]:
        n = max(n // SHRINK, 8192) + (n % 64)
        data = DATASETS[dname](n)
        ro, och = orc_compress(O, data, T, 5, shuffle, codec)
        r, out = pkg.decompress(och, n)
        assert r == n and np.array_equal(out, data), ("decode", codec, dname, T, shuffle, r)
        rc, ch = pkg.compress(data, T, 5, shuffle, codec.encode())
        assert rc > 0, ("encode", codec, dname, T, shuffle, rc)
        rr, o2 = orc_decompress(O, ch, n)
        assert rr == n and np.array_equal(o2, data), ("oracle reads GPU chunk", codec, dname, T, shuffle, rr)
        n_ok += 1
# BLOSC_MODE_CHECK_Z=1 (the round-3 switches of tests/test_gpu_modes.py): the entropy-coded formats as well - chunks written here are read by
# the oracle and by our own decoder (fused or stand-alone unshuffle, k_zstd_seq with either table form, zlib's queues)
if os.environ.get("BLOSC_MODE_CHECK_Z") == "1":
    for codec in ("zstd", "zlib"):
        for dname, T, shuffle, n in [("bench19", 8, 1, (4 << 20) + 24), ("linspace", 16, 1, 2 << 20), ("smallints", 2, 1, 1 << 20), ("randwalk", 8, 0, 1 << 20)]:
            n = max(n // SHRINK, 8192) + (n % 64)
            data = DATASETS[dname](n)
            rc, ch = pkg.compress(data, T, 5 if codec == "zlib" else 3, shuffle, codec.encode())
            assert rc > 0, ("encode", codec, dname, T, shuffle, rc)
            rr, o2 = orc_decompress(O, ch, n)
            assert rr == n and np.array_equal(o2, data), ("oracle reads GPU chunk", codec, dname, T, shuffle, rr)
            r, out = pkg.decompress(ch, n)
            assert r == n and np.array_equal(out, data), ("decode", codec, dname, T, shuffle, r)
            n_ok += 1
# a periodic plane followed by a match reaching back into it (the "materialise" path of the spans)
neb = 64 << 10
planes = []
rng = np.random.default_rng(5)
for j in range(8):
    head = rng.integers(0, 256, 16, dtype=np.uint8)
    p = np.empty(neb, np.uint8)
    p[:16] = head
    for k in range(16, neb - 4096):
        p[k] = p[k - 16]
    p[neb - 4096:] = p[100:4196] if j % 2 else rng.integers(0, 3, 4096, dtype=np.uint8)
    planes.append(p)
streams = []
for p in planes:
    outb = np.zeros(neb + 1024, np.uint8)
    k = O.orc_lz4_compress(ptr(p), neb, ptr(outb), neb + 1024, 1)
    assert 0 < k < neb
    streams.append(outb[:k].tobytes())
chunk = wrap_planes_as_chunk(streams, neb, 1)
want = np.stack(planes, 1).reshape(-1)
r, out = pkg.decompress(chunk, want.size)
assert r == want.size and np.array_equal(out, want), ("hand-built periodic planes", r)
print("modes ok", n_ok + 1)
