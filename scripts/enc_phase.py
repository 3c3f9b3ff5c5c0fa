"""Phase profile of the encoder (instrumented build): per byte-plane averages of the in-kernel counters."""
import ctypes as C, importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
mod.LIB_PATH = os.path.join(ROOT, "c-blosc_amd", "libblosc_amd_prof.so")
lib = mod.load()
nchunks = int(os.environ.get("CHUNKS", "128")); csz = 64 << 20
dname = os.environ.get("DATA", "bench19"); codec = os.environ.get("CODEC", "lz4").encode()
TS = int(os.environ.get("TYPESIZE", "8")); SHUF = int(os.environ.get("SHUFFLE", "1"))
dev = torch.device("cuda:0")
host = DATASETS[dname](csz)
src = torch.from_numpy(host).to(dev).unsqueeze(0).expand(nchunks, csz).contiguous()
comp = torch.empty((nchunks, csz + 16), dtype=torch.uint8, device=dev)
bc = mod.DeviceBatch([src[i].data_ptr() for i in range(nchunks)], [csz] * nchunks, [comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks)
bc.compress(TS, int(os.environ.get("CLEVEL", "5")), SHUF, codec, 0)
os.environ["BLOSC_AMD_ENC_PROFILE"] = "/tmp/encprof.bin"
lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
bc.compress(TS, int(os.environ.get("CLEVEL", "5")), SHUF, codec, 0)
lib.blosc_gpu_profile(0)
d = mod.profile_get("k_zstd_encode" if codec == b"zstd" else "k_encode_streams")
p = np.fromfile("/tmp/encprof.bin", np.uint32).reshape(-1, 16).astype(np.float64)
print(f"{dname} {codec.decode()}: kernel {d[0]/d[1]:.3f} ms (instrumented), streams {p.shape[0]}, ratio {csz/bc.results()[0]:.2f}")
zs = codec == b"zstd"
names = {0: "steps", 1: "nomatch", 6: "seqs", 2: "fwd_ext", 5: "bwd_tried", 7: "bwd>0", 4: "bwd>4", 3: "lit_mem", 8: "cyc_probe", 9: "cyc_cand", 10: "cyc_ext", 11: "cyc_emit", 12: "cyc_tail"}
NP = TS if SHUF == 1 else (TS if TS > 1 else 1)
for plane in range(NP):
    q = p[plane::NP].mean(axis=0)
    tot = q[8:13].sum() + (q[4] + q[5] if codec == b"zstd" else 0)
    print(f" plane {plane}: " + "  ".join(f"{names[i]}={q[i]:.0f}" for i in names) + f"  | total cyc {tot:.0f}  cyc/step {tot / max(q[0], 1):.0f}")
if zs: print(" Zstd: slot bwd>4 = cycles tail literals + offset values, bwd_tried = cycles sequences section")
print(" max stream total cycles:", p[:, 8:13].sum(axis=1).max(), " mean:", p[:, 8:13].sum(axis=1).mean())
