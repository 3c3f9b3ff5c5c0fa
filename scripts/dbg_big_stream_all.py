"""Probe (round 6): ONE stream beyond 256 MiB for every codec and filter - forced blocksize = the chunk, BLOSC_SPLITMODE=NEVER.
Written here and read by the reference, written by the reference and read here.  (The LZ4 writer's 28-bit position packing was found this way:
scripts/dbg_big_stream.py; this is the same question put to the other writers, the decoders and the fused filters.)
    python scripts/dbg_big_stream_all.py [codec ...]          FILTERS=0,1,2  MIB=266"""
import ctypes as C, importlib.util, os, sys, numpy as np, torch, time
ROOT = '/root/repo' if os.path.exists('/root/repo/tests') else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
torch.cuda.init(); lib = mod.load()
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
for L in (lib, R):
    L.blosc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
    L.blosc_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.blosc_set_compressor.argtypes = [C.c_char_p]
    L.blosc_set_blocksize.argtypes = [C.c_size_t]
codecs = sys.argv[1:] or ["blosclz", "lz4", "lz4hc", "zlib", "zstd"]
filters = [int(f) for f in os.environ.get("FILTERS", "0,1,2").split(",")]
mib = int(os.environ.get("MIB", "266"))
n = mib << 20
rng = np.random.default_rng(1)
head = rng.integers(0, 256, n - (8 << 20), dtype=np.uint8)
tail = np.ascontiguousarray(DATASETS["bench19"](8 << 20).reshape(-1, 8).T).reshape(-1)
plain = {0: np.concatenate([head, tail]), 1: DATASETS["bench19"](n), 2: DATASETS["bench19"](n)}      # filters on: every plane of the block has its matches, the last plane lies beyond 2^27 * 1.75
del head, tail
dst = np.empty(n + 16, np.uint8)
back = np.empty(n, np.uint8)
os.environ["BLOSC_SPLITMODE"] = "NEVER"
bad = 0
for codec in codecs:
    for f in filters:
        data = plain[f]
        cl = 1 if codec in ("zlib", "zstd", "lz4hc") else 5
        for writer, reader, wn, rn in ((lib, R, "here", "reference"), (R, lib, "reference", "here"), (lib, lib, "here", "here")):
            writer.blosc_init(); writer.blosc_set_compressor(codec.encode()); writer.blosc_set_blocksize(n)
            t0 = time.time()
            cb = writer.blosc_compress(cl, f, 8, n, data.ctypes.data, dst.ctypes.data, n + 16)
            t1 = time.time()
            writer.blosc_set_blocksize(0); writer.blosc_destroy()
            hdr = dst[:16].view("<u4")
            back[:] = 0
            reader.blosc_init()
            r = reader.blosc_decompress(dst.ctypes.data, back.ctypes.data, n)
            t2 = time.time()
            reader.blosc_destroy()
            ok = r == n and bool(np.array_equal(back, data))
            bad += not ok
            print(f"{codec:8s} filter {f} clevel {cl}: written {wn} ({t1 - t0:.1f} s) cbytes {cb} blocksize {hdr[2]} flags {dst[2]:#x}; read {rn} ({t2 - t1:.1f} s): {'ok' if ok else 'BAD r=' + str(r)}", flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
