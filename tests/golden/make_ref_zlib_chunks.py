#!/usr/bin/env python3
"""Writes tests/golden/ref_zlib_chunks.npz: Zlib chunks produced by the REAL reference (oracle/_ref, built from
/root/reference by oracle/Makefile), for pinning oracle/zlib_oracle.c where the reference is not available.
Run in the build container:  python tests/golden/make_ref_zlib_chunks.py"""
import ctypes as C, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS, ptr
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
out = {}; meta = []
k = 0
GEOM = ((300001, 8, 3, 1, 0), (1 << 19, 4, 5, 2, 0), (1 << 20, 8, 9, 1, 0), (70000, 1, 1, 0, 0), (1 << 19, 8, 7, 1, 65536))
for dname in ("bench19", "linspace", "smallints", "zeros", "randwalk", "random"):
    # nearly incompressible sets stay small: the fixture is committed
    for (n, T, clevel, shuffle, bs) in (GEOM if dname not in ("randwalk", "random") else ((150001, 8, 3, 1, 0), (70000, 4, 9, 2, 0))):
        data = DATASETS[dname](n)
        buf = np.empty(n + 16, np.uint8)
        r = R.blosc_compress_ctx(clevel, shuffle, T, n, ptr(data), ptr(buf), n + 16, b"zlib", bs, 1)
        assert r > 0
        out[f"c{k}"] = buf[:r].copy(); meta.append(f"{dname},{n},{T},{clevel},{shuffle},{bs}"); k += 1
out["meta"] = np.array(meta)
np.savez_compressed(os.path.join(HERE, "ref_zlib_chunks.npz"), **out)
print(k, "chunks,", sum(v.size for kk, v in out.items() if kk != "meta"), "bytes")
