"""PCIe-inclusive rate of the STOCK ABI (host buffers): blosc_compress / blosc_decompress on numpy arrays."""
import ctypes as C, importlib.util, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
torch.cuda.init()
lib = mod.load()
lib.blosc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
lib.blosc_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.blosc_init(); lib.blosc_set_compressor(b"lz4")
for mib in (64, 1024):
    n = mib << 20
    if n > 2**31 - 32: n = 2**31 - 64 - ((2**31 - 64) % 8)
    src = DATASETS["bench19"](n); dst = np.empty(n + 16, np.uint8); back = np.empty(n, np.uint8)
    cb = lib.blosc_compress(5, 1, 8, n, src.ctypes.data, dst.ctypes.data, n + 16)
    assert cb > 0
    reps = 5 if mib == 64 else 2
    t0 = time.perf_counter()
    for _ in range(reps): cb = lib.blosc_compress(5, 1, 8, n, src.ctypes.data, dst.ctypes.data, n + 16)
    t1 = time.perf_counter()
    for _ in range(reps): r = lib.blosc_decompress(dst.ctypes.data, back.ctypes.data, n)
    t2 = time.perf_counter()
    assert r == n and np.array_equal(back, src)
    print(f"host-buffer ABI, one {n / 2**20:.0f} MiB chunk: compress {n * reps / (t1 - t0) / 1e9:.2f} GB/s, decompress {n * reps / (t2 - t1) / 1e9:.2f} GB/s (pageable numpy memory, PCIe both ways included)")
