"""Phase profile of k_decode_blocks (instrumented build libblosc_amd_prof.so): per byte-plane averages of the
in-kernel counters over all persistent workgroups (see the slot list in k_decode_blocks.hip)."""
import ctypes as C, importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
mod.LIB_PATH = os.path.join(ROOT, "c-blosc_amd", "libblosc_amd_prof.so")
lib = mod.load()
nchunks = int(os.environ.get("CHUNKS", "128")); csz = 64 << 20
dname = os.environ.get("DATA", "bench19"); who = os.environ.get("WRITER", "stock")
host = DATASETS[dname](csz)
dev = torch.device("cuda:0")
if who == "stock":
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
    R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    tmp = np.empty(csz + 16, np.uint8)
    r = R.blosc_compress_ctx(5, 1, 8, csz, host.ctypes.data, tmp.ctypes.data, csz + 16, b"lz4", 0, 1)
    chunk = tmp[:r].copy()
else:
    r, chunk = mod.compress(host, 8, 5, 1, b"lz4")
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev)
comp[:, :r].copy_(torch.from_numpy(chunk).to(dev).unsqueeze(0).expand(nchunks, r))
back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
bd = mod.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
bd.decompress()
os.environ["BLOSC_AMD_BD_PROFILE"] = "/tmp/bdprof.bin"
lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
bd.decompress()
lib.blosc_gpu_profile(0)
d = mod.profile_get("k_decode_blocks")
ok = bool((back[0] == torch.from_numpy(host).to(dev)).all())
import glob
fn = "/tmp/bdprof.bin.T8W" + os.environ.get("W", "4")
T = int(fn[-1])
p = np.fromfile(fn, np.uint32).reshape(-1, 16).astype(np.float64)
print(f"{who} {dname}: k_decode_blocks {d[0]/max(d[1],1):.3f} ms (instrumented), waves {p.shape[0]}, ok={ok}")
names = {0: "steps", 1: "step_seq", 2: "rest", 3: "scalar_tok", 5: "far_chunks", 8: "cyc_decode", 6: "cyc_window", 7: "cyc_parse", 10: "cyc_step_to_rest",
         11: "cyc_rest", 12: "cyc_far", 13: "cyc_scalar_match", 9: "cyc_barrier", 4: "cyc_writeout", 14: "cyc_kernel", 15: "blocks"}
for plane in range(T):
    q = p[plane::T].mean(axis=0)
    nb = max(q[15], 1)
    print(f" wave {plane} per block: " + "  ".join(f"{names[i]}={q[i] / nb:.0f}" for i in names if i != 15) + f"  blocks/WG {q[15]:.1f}")
