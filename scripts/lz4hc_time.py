"""Kernel time of the "lz4hc" encoder (k_lz4hc_encode) next to the plain LZ4 match finder under the same name
(BLOSC_AMD_LZ4HC=0 -> k_encode_streams), BASELINE geometry: 64 MiB chunks of bench19, typesize 8, byte shuffle, clevel 9.
No torch: host buffers through blosc_gpu_compress_batch_host, times from the library's own event pairs (blosc_gpu_profile)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS  # noqa: E402

pkg = importlib.import_module("c-blosc_amd")
lib = pkg.load()
n, csz = int(os.environ.get("NCHUNKS", "8")), 64 << 20
data = DATASETS[os.environ.get("DATASET", "bench19")](csz)
srcs = [data.copy() for _ in range(n)]
dsts = [np.empty(csz + 16, np.uint8) for _ in range(n)]
sp = (C.c_void_p * n)(*[a.ctypes.data for a in srcs]); dp = (C.c_void_p * n)(*[a.ctypes.data for a in dsts])
ssz = (C.c_size_t * n)(*[csz] * n); dsz = (C.c_size_t * n)(*[csz + 16] * n); res = (C.c_int * n)()
for mode, kernel in (("1", b"k_lz4hc_encode"), ("0", b"k_encode_streams")):
    os.environ["BLOSC_AMD_LZ4HC"] = mode
    for rep in range(3):
        lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
        rc = lib.blosc_gpu_compress_batch_host(9, 1, 8, b"lz4hc", 0, n, sp, ssz, dp, dsz, res)
        ms, k = C.c_double(0), C.c_int(0)
        lib.blosc_gpu_profile_get(kernel, C.byref(ms), C.byref(k))
        lib.blosc_gpu_profile(0)
        assert rc == 0 and all(r > 0 for r in res), (rc, list(res))
    print(f"BLOSC_AMD_LZ4HC={mode}: {kernel.decode()} {ms.value:.2f} ms for {n} x 64 MiB = {n * csz / ms.value / 1e6:.1f} GB/s, ratio {csz / res[0]:.2f}")
del os.environ["BLOSC_AMD_LZ4HC"]
rc = lib.blosc_gpu_compress_batch_host(9, 1, 8, b"lz4hc", 0, n, sp, ssz, dp, dsz, res)
print(f"environment unset: ratio {csz / res[0]:.2f}")
