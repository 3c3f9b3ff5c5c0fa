"""scripts/placement_skew.py + simple probe kernels (scripts/micro/lat_probe.hip) run over the SAME arena right after the timed decodes: do dependent
256-byte reads, row writes / reads that wait for every row, or a plain copy see the decode kernel's slow / fast state of that allocation?"""
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
P = C.CDLL(os.path.join(ROOT, "scripts", "micro", "liblat_probe.so"))
P.lat_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_float)]
def probe(base, mode, iters):
    best = 1e9
    for _ in range(3):
        ms = C.c_float(0)
        assert P.lat_probe(base + (1 << 30), 6 << 30, mode, iters, C.byref(ms)) == 0
        best = min(best, ms.value)
    return best
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
    base = int(a.split()[0], 16)
    pr = [probe(base, 0, 1500), probe(base, 1, 4), probe(base, 2, 4), probe(base, 3, 0)]
    rows.append((skew, f"{base:#x}", d[0] / d[1], pr))
os.dup2(_saved, 2)
print("probes: chase = 4096 waves x 1500 dependent 256-byte reads; rowwr / rowrd = 2 GiB of 1 KiB rows, each waited for; copy = 3 GiB -> 3 GiB")
for skew, a, ms, pr in rows: print(f"skew {skew:8d} KiB  arena {a}  decode {ms:.3f} ms   chase {pr[0]:.3f}  rowwr {pr[1]:.3f}  rowrd {pr[2]:.3f}  copy {pr[3]:.3f} ms")
import numpy as _np
dd = _np.array([r[2] for r in rows])
for i, nm in enumerate(("chase", "rowwr", "rowrd", "copy")):
    pp = _np.array([r[3][i] for r in rows]); print(f"correlation of decode time with {nm}: {_np.corrcoef(dd, pp)[0, 1]:+.2f}   (probe spread {pp.min():.3f} .. {pp.max():.3f} ms)")
