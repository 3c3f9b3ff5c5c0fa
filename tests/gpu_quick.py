"""Fast GPU sanity script (no pytest, no torch): prints one line per check.  Used as the first
thing in a gpurun call so that a broken kernel is visible before the long test run."""
import ctypes as C
import glob
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS, orc_compress, orc_decompress, ptr  # noqa: E402

spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
L = pkg.load()
O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
sz, i, vp = C.c_size_t, C.c_int, C.c_void_p
O.orc_compress.argtypes = [i, i, sz, sz, vp, vp, sz, i, sz, i]
O.orc_decompress.argtypes = [vp, vp, sz]
for f in (O.orc_shuffle, O.orc_unshuffle, O.orc_bitshuffle, O.orc_bitunshuffle):
    f.argtypes = [sz, sz, vp, vp]

rng = np.random.default_rng(0)
for T, n in [(8, 1 << 20), (4, 1 << 19), (2, 4096), (16, 65536), (3, 30000), (255, 255 * 300)]:
    src = rng.integers(0, 256, n, dtype=np.uint8)
    w = np.zeros(n, np.uint8); g = np.zeros(n, np.uint8)
    O.orc_shuffle(T, n, ptr(src), ptr(w)); L.blosc_internal_shuffle(T, n, ptr(src), ptr(g))
    a = np.array_equal(w, g)
    O.orc_bitshuffle(T, n, ptr(src), ptr(w)); L.blosc_internal_bitshuffle(T, n, ptr(src), ptr(g), None)
    b = np.array_equal(w, g)
    back = np.zeros(n, np.uint8); L.blosc_internal_bitunshuffle(T, n, ptr(g), ptr(back), None)
    c = np.array_equal(back, src)
    L.blosc_internal_shuffle(T, n, ptr(src), ptr(g)); L.blosc_internal_unshuffle(T, n, ptr(g), ptr(back), None) if False else L.blosc_internal_unshuffle(T, n, ptr(g), ptr(back))
    d = np.array_equal(back, src)
    print(f"filters T={T} n={n}: shuffle {a} bitshuffle {b} bitunshuffle {c} unshuffle {d}", flush=True)

exp = np.arange(10**6, dtype="<i4")
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "compat", "*.cdata"))):
    ch = np.fromfile(f, np.uint8)
    r, out = pkg.decompress(ch, 4000000)
    ok = r == 4000000 and np.array_equal(out.view("<i4"), exp)
    print(f"compat {os.path.basename(f):40s} -> {r} {'OK' if ok else ''}", flush=True)

for codec in ["lz4", "blosclz"]:
    for dname in ["bench19", "randwalk", "zeros", "random", "linspace"]:
        for shuffle in [1, 2, 0]:
            n = (1 << 22) + 24
            data = DATASETS[dname](n)
            ro, ochunk = orc_compress(O, data, 8, 5, shuffle, codec)
            r, out = pkg.decompress(ochunk, n)
            dec_ok = r == n and np.array_equal(out, data)
            t0 = time.perf_counter()
            rc, chunk = pkg.compress(data, 8, 5, shuffle, codec.encode())
            t1 = time.perf_counter()
            enc_ok = False; rr = -99
            if rc > 0:
                rr, o2 = orc_decompress(O, chunk, n)
                enc_ok = rr == n and np.array_equal(o2, data)
            print(f"{codec:8s} {dname:9s} shuffle={shuffle}: decode(oracle chunk) {dec_ok} [{r}]  encode->oracle {enc_ok} [{rc},{rr}]"
                  f"  ratio gpu {n / max(rc, 1):7.2f} ref {n / ro:7.2f}  ({(t1 - t0) * 1e3:.1f} ms host call)", flush=True)
print("quick done", flush=True)
