"""Same-process A/B of the decompress direction: several builds of the library loaded side by side (ctypes keeps their globals apart),
the SAME device buffers, the builds taking turns round after round - so that buffer placement, clocks and whatever else differs between
two processes on one box cancels out.  Reference-written chunks (the drop-in direction).
    python scripts/dec_ab.py libA.so libB.so ...        env: DATA=bench19 SHUFFLE=1 TYPESIZE=8 CODEC=lz4 CLEVEL=5 CHUNKS=128 ROUNDS=5"""
import ctypes as C, importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS


def load(path, tag):
    spec = importlib.util.spec_from_file_location("c_blosc_amd_" + tag, os.path.join(ROOT, "c-blosc_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    mod.LIB_PATH = os.path.abspath(path)
    mod.load()
    return mod


libs = sys.argv[1:] or [os.path.join(ROOT, "c-blosc_amd", "libblosc_amd.so")]
mods = [load(p, str(k)) for k, p in enumerate(libs)]
nchunks = int(os.environ.get("CHUNKS", "128")); csz = 64 << 20
rounds = int(os.environ.get("ROUNDS", "5"))
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
dev = torch.device("cuda:0")
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev)
back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
names = ["k_decode_plan", "k_decode_streams", "k_unshuffle", "k_bitunshuffle", "k_zstd_streams", "k_zlib_streams"]
zparts = ["k_zstd_entropy", "k_zstd_seq", "k_zstd_exec"]      # (the experimental builds with the sliced Zstd pipeline of round 4 bracketed them as k_zstd_pipeline: fork ... join)
def decode_ms(m, calls):
    t = sum(m.profile_get(n)[0] / calls for n in names if m.profile_get(n)[1])
    if m.profile_get("k_zstd_pipeline")[1]: return t + m.profile_get("k_zstd_pipeline")[0] / calls
    return t + sum(m.profile_get(n)[0] / calls for n in zparts if m.profile_get(n)[1])
for spec in os.environ.get("DECSETS", "bench19:1:8").split():
    dname, sh, ts = spec.split(":"); sh, ts = int(sh), int(ts)
    host = DATASETS[dname](csz)
    tmp = np.empty(csz + 16, np.uint8)
    r = R.blosc_compress_ctx(int(os.environ.get("CLEVEL", "5")), sh, ts, csz, host.ctypes.data, tmp.ctypes.data, csz + 16, os.environ.get("CODEC", "lz4").encode(), 0, 8)
    comp[:, :r].copy_(torch.from_numpy(tmp[:r].copy()).to(dev).unsqueeze(0).expand(nchunks, r))
    want = torch.from_numpy(host).to(dev)
    if os.environ.get("OWN"):      # OWN=1: the chunks the FIRST library writes instead of the reference's (every build decodes the same bytes)
        back[:] = want
        eb = mods[0].DeviceBatch([back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks, [comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks)
        assert eb.compress(ts, int(os.environ.get("CLEVEL", "5")), sh, os.environ.get("CODEC", "lz4").encode(), 0) == 0
        r = eb.results()[0]
    batches = [m.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks) for m in mods]
    res = [[] for _ in mods]
    for m, b in zip(mods, batches):
        back.zero_(); b.decompress(); b.decompress()
        assert os.environ.get("NOCHECK") or (b.results() == [csz] * nchunks and bool((back[0] == want).all()) and bool((back[-1] == want).all())), m.LIB_PATH
    for _ in range(rounds):
        for k, (m, b) in enumerate(zip(mods, batches)):
            L = m.load()
            L.blosc_gpu_profile(1); L.blosc_gpu_profile_reset()
            for _ in range(3): b.decompress()
            L.blosc_gpu_profile(0)
            res[k].append(decode_ms(m, 3))
    print(f"{dname} shuffle={sh} T={ts} ratio={csz / r:.1f}: " + "   ".join(f"{os.path.basename(p)} {np.median(v):.3f} ms (min {min(v):.3f})" for p, v in zip(libs, res)), flush=True)
