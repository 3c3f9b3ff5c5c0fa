"""Probe (round 6): ONE LZ4 stream beyond 256 MiB - forced blocksize = the chunk, BLOSC_SPLITMODE=NEVER - written here, read by the reference."""
import ctypes as C, importlib.util, os, sys, numpy as np, torch, time
ROOT='/root/repo' if os.path.exists('/root/repo/tests') else os.getcwd()
sys.path.insert(0, os.path.join(ROOT,'tests'))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
torch.cuda.init(); lib = mod.load()
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
R.blosc_decompress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
lib.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
rng = np.random.default_rng(1)
head = rng.integers(0, 256, 258 << 20, dtype=np.uint8)
tail = DATASETS["bench19"](8 << 20)
tail = np.ascontiguousarray(tail.reshape(-1, 8).T).reshape(-1)      # shuffled planes as one byte stream: matches at short distances
data = np.concatenate([head, tail]); n = data.size
dst = np.empty(n + 16, np.uint8)
os.environ["BLOSC_SPLITMODE"] = "NEVER"
lib.blosc_init(); lib.blosc_set_compressor(b"lz4"); lib.blosc_set_blocksize(n)
lib.blosc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
for cl in (5, 9):
    t0 = time.time()
    cb = lib.blosc_compress(cl, 0, 8, n, data.ctypes.data, dst.ctypes.data, n + 16)
    t1 = time.time()
    back = np.empty(n, np.uint8)
    r = R.blosc_decompress_ctx(dst.ctypes.data, back.ctypes.data, n, 4)
    print("clevel", cl, "cbytes", cb, "of", n, f"{t1 - t0:.1f} s", "reference reads it:", r == n and bool(np.array_equal(back, data)), flush=True)
    hdr = dst[:16].view("<u4")
    print("   header: nbytes", hdr[1], "blocksize", hdr[2], "cbytes", hdr[3], "flags", dst[2])
