"""ONE instance of the library; its arenas are given back and re-allocated with a different offset (BLOSC_AMD_ARENA_SKEW_KIB) between the address
hipMalloc returns and the address the engine uses: decode time of reference-written config-2 chunks per offset.  Is the 7 - 10 % spread between
instances / processes a function of the arena's ADDRESS?"""
import ctypes as C, importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
os.environ["BLOSC_AMD_DEBUG"] = "1"
os.environ["BLOSC_AMD_DEBUG_COST"] = "1"
import tempfile
errf = tempfile.mktemp(prefix="bamd_skew_"); _fd = os.open(errf, os.O_WRONLY | os.O_CREAT | os.O_TRUNC); _saved = os.dup(2); os.dup2(_fd, 2)
def arenas(state=[0]):
    C.CDLL(None).fflush(None)
    with open(errf) as fh:
        fh.seek(state[0]); t = fh.read(); state[0] = fh.tell()
    costs = [ln.split("costs:", 1)[1].split()[:8] for ln in t.splitlines() if "plane costs" in ln]
    return " ".join(ln.split("arena", 1)[1].split(",")[0].strip() for ln in t.splitlines() if "arena" in ln and " 1 MiB" not in ln) + (" costs(last) " + ",".join(costs[-1]) if costs else "")
L = mod.load()
nchunks, csz = 128, 64 << 20
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
host = DATASETS["bench19"](csz); tmp = np.empty(csz + 16, np.uint8)
r = R.blosc_compress_ctx(5, 1, 8, csz, host.ctypes.data, tmp.ctypes.data, csz + 16, b"lz4", 0, 8)
dev = torch.device("cuda:0")
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev); back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
comp[:, :r].copy_(torch.from_numpy(tmp[:r].copy()).to(dev).unsqueeze(0).expand(nchunks, r))
bd = mod.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
print(f"buffers: comp {comp.data_ptr():#x} back {back.data_ptr():#x}")
rng = np.random.default_rng(5)
skews = [int(x) for x in os.environ.get("SKEWS", "0 4 8 16 32 64 128 256 512 1024 2048 2052 4096 4100 0 4").split()]      # KiB; "0 0 0 ..." repeats one placement
rows = []
for skew in skews:
    L.blosc_init(); L.blosc_destroy(); arenas()
    os.environ["BLOSC_AMD_ARENA_SKEW_KIB"] = str(skew)
    assert bd.decompress() == 0 and bd.decompress() == 0 and bd.decompress() == 0
    a = arenas()
    L.blosc_gpu_profile(1); L.blosc_gpu_profile_reset()
    for _ in range(4): bd.decompress()
    L.blosc_gpu_profile(0)
    d = mod.profile_get("k_decode_streams")
    rows.append((skew, a + " | timed: " + arenas(), d[0] / d[1]))
os.dup2(_saved, 2)
for skew, a, ms in rows: print(f"skew {skew:8d} KiB  arena {a}  decode {ms:.3f} ms")
