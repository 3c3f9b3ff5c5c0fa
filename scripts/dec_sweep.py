"""Decode-only timing of stock (reference-written) chunks for occupancy experiments."""
import ctypes as C, importlib.util, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
lib = mod.load()
nchunks = int(os.environ.get("CHUNKS", "128")); csz = 64 << 20
dname = os.environ.get("DATA", "bench19")
host = DATASETS[dname](csz)
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
tmp = np.empty(csz + 16, np.uint8)
r = R.blosc_compress_ctx(int(os.environ.get("CLEVEL", "5")), int(os.environ.get("SHUFFLE", "1")), int(os.environ.get("TYPESIZE", "8")), csz, host.ctypes.data, tmp.ctypes.data, csz + 16, os.environ.get("CODEC", "lz4").encode(), 0, 1)
dev = torch.device("cuda:0")
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev)
comp[:, :r].copy_(torch.from_numpy(tmp[:r].copy()).to(dev).unsqueeze(0).expand(nchunks, r))
back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
bd = mod.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
bd.decompress()
lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
for _ in range(3): bd.decompress()
lib.blosc_gpu_profile(0)
if not os.environ.get("NOCHECK"): assert bd.results() == [csz] * nchunks      # NOCHECK=1: timing experiments with builds that leave work out on purpose
ok = bool((back[0] == torch.from_numpy(host).to(dev)).all()) and bool((back[-1] == torch.from_numpy(host).to(dev)).all())
names = ["k_decode_plan", "k_decode_streams", "k_unshuffle", "k_bitunshuffle", "k_zstd_entropy", "k_zstd_seq", "k_zstd_exec", "k_zstd_streams", "k_zlib_streams"]
tot = 0.0; parts = []
for k in names:
    ms, cnt = mod.profile_get(k)
    if cnt: parts.append(f"{k} {ms / 3:.3f}"); tot += ms / 3
print(f"data={dname} chunks={nchunks} ratio={csz/r:.1f}: decompress kernels {tot:.3f} ms/call  [" + "  ".join(parts) + f"]  ok={ok}", flush=True)
