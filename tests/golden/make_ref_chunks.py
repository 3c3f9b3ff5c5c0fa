"""Generates tests/golden/ref_chunks.npz with the REAL reference (oracle/_ref/libblosc_ref.so, built
from /root/reference by oracle/Makefile).  Run in the dev container:  python tests/golden/make_ref_chunks.py

Each entry is a chunk written by stock blosc_compress_ctx for a small deterministic input that
tests/helpers.py can regenerate (dataset name + nbytes), including writers this repo does not
implement (lz4hc encoder; nthreads=4, whose block order in the payload is completion order,
blosc/blosc.c:1845-1860) so that the GPU decoder is checked on what stock c-blosc really emits."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import DATASETS, ptr  # noqa: E402

R = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libblosc_ref.so"))
R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
R.blosc_set_splitmode.argtypes = [C.c_int]

cases = []
for cname in ["blosclz", "lz4", "lz4hc"]:
    for shuffle in [0, 1, 2]:
        for (dname, n, T) in [("bench19", 300001, 8), ("bench19", (1 << 20) + 4, 4), ("randwalk", 40000 + 24, 8),
                              ("smallints", 60000, 4), ("arange", 400000, 4), ("zeros", 70000, 2), ("bench19", 40000, 16)]:
            if shuffle == 0 and n > 100000:
                continue          # unfiltered bench19 barely compresses: keep the fixture file small
            for clevel, nthreads in [(5, 1), (9, 4)]:
                cases.append((cname, shuffle, dname, n, T, clevel, nthreads, 0, 4))
# split modes and forced block sizes
for sm in [1, 2, 3]:
    cases.append(("lz4", 1, "bench19", 300001, 8, 5, 1, 0, sm))
    cases.append(("blosclz", 1, "bench19", 300001, 4, 5, 1, 16384, sm))

out = {}
meta = []
for k, (cname, shuffle, dname, n, T, clevel, nth, bs, sm) in enumerate(cases):
    data = DATASETS[dname](n)
    buf = np.zeros(n + 16, np.uint8)
    R.blosc_set_splitmode(sm)
    r = R.blosc_compress_ctx(clevel, shuffle, T, n, ptr(data), ptr(buf), n + 16, cname.encode(), bs, nth)
    assert r > 0
    out[f"c{k}"] = buf[:r].copy()
    meta.append(f"{cname},{shuffle},{dname},{n},{T},{clevel},{nth},{bs},{sm}")
R.blosc_set_splitmode(4)
out["meta"] = np.array(meta)
np.savez_compressed(os.path.join(HERE, "ref_chunks.npz"), **out)
print(len(cases), "chunks,", sum(v.size for k, v in out.items() if k != "meta"), "bytes")
