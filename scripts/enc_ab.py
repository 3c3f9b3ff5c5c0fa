"""Same-process A/B of the compress direction (see dec_ab.py): several builds side by side, the same device buffers, taking turns.
    python scripts/enc_ab.py libA.so libB.so ...      env: ENCSETS="bench19:1:8 ..." CODEC=lz4 CLEVEL=5 CHUNKS=128 ROUNDS=5"""
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
codec = os.environ.get("CODEC", "lz4").encode(); clevel = int(os.environ.get("CLEVEL", "5"))
dev = torch.device("cuda:0")
src = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev)
names = ["k_shuffle", "k_bitshuffle", "k_encode_streams", "k_lz4hc_encode", "k_zstd_encode", "k_zlib_encode", "k_chunk_scan", "k_chunk_compact"]
for spec in os.environ.get("ENCSETS", "bench19:1:8").split():
    dname, sh, ts = spec.split(":"); sh, ts = int(sh), int(ts)
    host = DATASETS[dname](csz)
    src.copy_(torch.from_numpy(host).to(dev).unsqueeze(0).expand(nchunks, csz))
    batches = [m.DeviceBatch([src[i].data_ptr() for i in range(nchunks)], [csz] * nchunks, [comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks) for m in mods]
    res = [[] for _ in mods]; ratio = []
    for m, b in zip(mods, batches):
        assert b.compress(ts, clevel, sh, codec, 0) == 0 and b.compress(ts, clevel, sh, codec, 0) == 0
        ratio.append(csz / b.results()[0])
    for _ in range(rounds):
        for k, (m, b) in enumerate(zip(mods, batches)):
            L = m.load()
            L.blosc_gpu_profile(1); L.blosc_gpu_profile_reset()
            for _ in range(3): b.compress(ts, clevel, sh, codec, 0)
            L.blosc_gpu_profile(0)
            res[k].append(sum(m.profile_get(n)[0] / 3 for n in names if m.profile_get(n)[1]))
    print(f"{dname} shuffle={sh} T={ts} {codec.decode()} cl{clevel}: " + "   ".join(f"{os.path.basename(p)} {np.median(v):.3f} ms (min {min(v):.3f}, ratio {r:.2f})" for p, v, r in zip(libs, res, ratio)), flush=True)
