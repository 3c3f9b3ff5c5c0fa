"""Phase profile of the LZ4 decoder (instrumented build): per byte-plane averages of the in-kernel counters."""
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
TS = int(os.environ.get("TYPESIZE", "8")); SHUF = int(os.environ.get("SHUFFLE", "1"))
host = DATASETS[dname](csz)
dev = torch.device("cuda:0")
if who == "stock":
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
    R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    tmp = np.empty(csz + 16, np.uint8)
    r = R.blosc_compress_ctx(5, SHUF, TS, csz, host.ctypes.data, tmp.ctypes.data, csz + 16, b"lz4", 0, 1)
    chunk = tmp[:r].copy()
else:
    r, chunk = mod.compress(host, TS, 5, SHUF, b"lz4")
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev)
comp[:, :r].copy_(torch.from_numpy(chunk).to(dev).unsqueeze(0).expand(nchunks, r))
back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
bd = mod.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
bd.decompress()
os.environ["BLOSC_AMD_DEC_PROFILE"] = "/tmp/decprof.bin"
lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
bd.decompress()
lib.blosc_gpu_profile(0)
d = mod.profile_get("k_decode_streams")
p = np.fromfile("/tmp/decprof.bin", np.uint32).reshape(-1, 16).astype(np.float64)
print(f"{who} {dname}: kernel {d[0]/d[1]:.3f} ms (instrumented), streams {p.shape[0]}")
names = {0: "batches", 1: "batch_seq", 2: "others", 3: "scalar_seq", 8: "cyc_parse", 9: "cyc_walk", 10: "cyc_lit+pieces", 11: "cyc_others", 12: "cyc_scalar", 13: "cyc_rest"}
for plane in range(TS):
    q = p[plane::TS].mean(axis=0)
    tot = q[8:14].sum()
    print(f" plane {plane}: " + "  ".join(f"{names[i]}={q[i]:.0f}" for i in names) + f"  | total cyc {tot:.0f}  cyc/seq {tot / max(q[1] + q[3], 1):.0f}")
allm = p.mean(axis=0); print(" max stream total cycles:", p[:, 8:14].sum(axis=1).max(), " mean:", p[:, 8:14].sum(axis=1).mean())
un = p[p[:, 7] > 0][:, 6]
if un.size: print(f" fused unshuffle: {un.size} blocks, mean {un.mean():.0f} cycles per block (max {un.max():.0f}); per stream {un.sum() / p.shape[0]:.0f} - streams' own mean {p[:, 8:14].sum(axis=1).mean():.0f}")


# concurrency per XCD (s_memtime counters are per XCD): mean number of streams in flight
raw = np.fromfile("/tmp/decprof.bin", np.uint32).reshape(-1, 16)
hw = raw[:, 4]; xcc = raw[:, 5] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3; wave = hw & 0xf
print(" wave slot ids seen:", np.unique(wave), " simd:", np.unique(simd))
for x in range(8):
    m = xcc == x
    st = raw[m, 14].astype(np.int64); en = raw[m, 15].astype(np.int64)
    t0 = st.min(); st -= t0; en -= t0; en[en < st] += 1 << 32
    span = en.max()
    key = (se[m].astype(np.int64) << 8) | (sh[m].astype(np.int64) << 4) | cu[m]
    ncu = np.unique(key).size
    tl = [int(((st <= f * span) & (en > f * span)).sum()) for f in (0.1, 0.3, 0.5, 0.7, 0.9, 0.97)]
    print(f" xcc {x}: streams {m.sum()} CUs {ncu} span {span * 64 / 1e6:.2f} Mticks  mean in flight {(en - st).sum() / span:.0f} ({(en - st).sum() / span / ncu:.1f}/CU)  timeline {tl}")
