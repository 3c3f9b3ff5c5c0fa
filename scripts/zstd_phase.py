"""Phase profile of k_zstd_streams (instrumented build): mean cycles per frame and phase, reference-written chunks."""
import ctypes as C, importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
mod.LIB_PATH = os.path.join(ROOT, "c-blosc_amd", "libblosc_amd_prof.so")
lib = mod.load()
nchunks = int(os.environ.get("CHUNKS", "32")); csz = 64 << 20
dname = os.environ.get("DATA", "bench19")
host = DATASETS[dname](csz)
dev = torch.device("cuda:0")
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
tmp = np.empty(csz + 16, np.uint8)
r = R.blosc_compress_ctx(3, 1, 8, csz, host.ctypes.data, tmp.ctypes.data, csz + 16, b"zstd", 0, 8)
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev)
comp[:, :r].copy_(torch.from_numpy(tmp[:r].copy()).to(dev).unsqueeze(0).expand(nchunks, r))
back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
bd = mod.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
bd.decompress()
os.environ["BLOSC_AMD_ZSTD_PROFILE"] = "/tmp/zprof.bin"
lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
bd.decompress()
lib.blosc_gpu_profile(0)
d = mod.profile_get("k_zstd_streams")
ok = bool((back[0] == torch.from_numpy(host).to(dev)).all())
p = np.fromfile("/tmp/zprof.bin", np.uint32).reshape(-1, 16).astype(np.float64)
p = p[p[:, 10] > 0]
q = p.mean(axis=0)
names = ["lit hdr + huf table", "huf streams", "seq tables", "seq decode (lane 0)", "seq execute", "rest"]
print(f"{dname} ratio {csz/r:.1f}: k_zstd_streams {d[0]/max(d[1],1):.2f} ms for {nchunks} chunks, frames {p.shape[0]}, ok={ok}")
print("  per frame: sequences %.0f  literals %.0f  blocks %.1f" % (q[8], q[9], q[10]))
tot = q[:6].sum()
for i, nme in enumerate(names): print(f"  {nme:24s} {q[i]:10.0f} cycles  {100*q[i]/tot:5.1f} %")
print(f"  total {tot:.0f} cycles per frame")
