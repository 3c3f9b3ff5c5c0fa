"""PCIe-inclusive rate of the stock host-buffer ABI from several threads at once (blosc_compress_ctx / blosc_decompress_ctx on numpy arrays):
what the context pool of DESIGN.md 4 buys.  Run once per BLOSC_AMD_CONTEXTS value (the pool size is read once per process):
    BLOSC_AMD_CONTEXTS=1 python scripts/host_abi_threads.py ; BLOSC_AMD_CONTEXTS=4 python scripts/host_abi_threads.py"""
import ctypes as C, importlib.util, os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
torch.cuda.init()
lib = mod.load()
n = int(os.environ.get("MIB", "64")) << 20
reps = int(os.environ.get("REPS", "6"))
src0 = DATASETS["bench19"](n)


def worker(bufs, out):
    src, dst, back = bufs
    cb = 0
    for _ in range(reps):
        cb = lib.blosc_compress_ctx(5, 1, 8, n, src.ctypes.data, dst.ctypes.data, n + 16, b"lz4", 0, 1)
        r = lib.blosc_decompress_ctx(dst.ctypes.data, back.ctypes.data, n, 1)
        assert cb > 0 and r == n
    out.append(bool(np.array_equal(back, src)))


for nt in (1, 2, 4, 8):
    bufs = [(src0.copy(), np.empty(n + 16, np.uint8), np.empty(n, np.uint8)) for _ in range(nt)]
    ok = []
    worker(bufs[0], ok)                                   # arenas, first touch of the buffers
    ok = []
    th = [threading.Thread(target=worker, args=(bufs[t], ok)) for t in range(nt)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"contexts={os.environ.get('BLOSC_AMD_CONTEXTS', '4 (default)')} threads={nt}: {nt * reps * n / dt / 1e9:6.2f} GB/s of round trips "
          f"(compress_ctx + decompress_ctx of one {n >> 20} MiB host chunk each, PCIe both ways, pageable memory)  ok={all(ok) and len(ok) == nt}", flush=True)
