"""The decode kernel's two states: does a decode that never touches the arena's plane-major scratch see them?  Per arena offset (a
re-allocation) two batches are timed on the SAME arena: the reference's config-2 chunks (split byte-shuffled blocks: planes through the scratch, hand-off,
fused unshuffle) and reference-written chunks of the same bytes ALREADY shuffled, compressed with doshuffle = 0 / typesize 1 (one stream per block,
decoded straight into the destination: of the arena only the small tables - stream / block descriptors, queues, tickets, status - are used)."""
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
pre = np.ascontiguousarray(host.reshape(-1, 8).T).reshape(-1)           # the same bytes plane-major: what config 2's streams hold
tmp2 = np.empty(csz + 16, np.uint8)
r2 = R.blosc_compress_ctx(5, 0, 1, csz, pre.ctypes.data, tmp2.ctypes.data, csz + 16, b"lz4", 0, 8)
comp2 = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev)
comp2[:, :r2].copy_(torch.from_numpy(tmp2[:r2].copy()).to(dev).unsqueeze(0).expand(nchunks, r2))
bd2 = mod.DeviceBatch([comp2[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
print(f"config 2: {r} bytes per chunk; pre-shuffled + noshuffle: {r2} bytes per chunk")
bd = mod.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
print(f"buffers: comp {comp.data_ptr():#x} back {back.data_ptr():#x}")
rng = np.random.default_rng(5)
skews = [int(x) for x in os.environ.get("SKEWS", "0 4 8 16 32 64 128 256 512 1024 2048 2052 4096 4100 0 4").split()]      # KiB; "0 0 0 ..." repeats one placement
rows = []
def timed(b):
    assert b.decompress() == 0
    L.blosc_gpu_profile(1); L.blosc_gpu_profile_reset()
    for _ in range(4): b.decompress()
    L.blosc_gpu_profile(0)
    d = mod.profile_get("k_decode_streams"); return d[0] / d[1]
for skew in skews:
    L.blosc_init(); L.blosc_destroy(); arenas()
    os.environ["BLOSC_AMD_ARENA_SKEW_KIB"] = str(skew)
    assert bd.decompress() == 0 and bd.decompress() == 0
    a = arenas().split()[0]
    t1 = timed(bd); t2 = timed(bd2); t3 = timed(bd); t4 = timed(bd2)
    assert "0x" not in arenas().split(" costs")[0], "the arena was re-allocated between the two batches"
    rows.append((skew, a, t1, t2, t3, t4))
os.dup2(_saved, 2)
for skew, a, t1, t2, t3, t4 in rows: print(f"skew {skew:6d} KiB arena {a}  config 2: {t1:.3f} {t3:.3f} ms   no scratch: {t2:.3f} {t4:.3f} ms")
x = np.array([(r[2] + r[4]) / 2 for r in rows]); y = np.array([(r[3] + r[5]) / 2 for r in rows])
print(f"correlation of the two over the offsets: {np.corrcoef(x, y)[0, 1]:+.2f}; no-scratch spread {y.min():.3f} .. {y.max():.3f} ms")
