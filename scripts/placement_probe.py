"""Why do instances of ONE build differ by 7 - 10 % inside one process (and fresh processes from box to box)?  N copies of the same library, each with
its own arenas (BLOSC_AMD_DEBUG=1 prints their addresses), the same input / output buffers, taking turns: decode time of reference-written config-2
chunks per instance next to the addresses of its arenas.   python scripts/placement_probe.py [ncopies=8]"""
import ctypes as C, importlib.util, os, shutil, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
os.environ["BLOSC_AMD_DEBUG"] = "1"
ncopies = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tmpd = tempfile.mkdtemp(prefix="bamd_probe_")
errf = os.path.join(tmpd, "stderr.txt")
fd = os.open(errf, os.O_WRONLY | os.O_CREAT | os.O_TRUNC); saved = os.dup(2); os.dup2(fd, 2)
def new_lines(state=[0]):
    C.CDLL(None).fflush(None)
    with open(errf) as fh:
        fh.seek(state[0]); t = fh.read(); state[0] = fh.tell()
    return [ln for ln in t.splitlines() if "arena" in ln]
def load(k):
    dst = os.path.join(tmpd, f"libcopy{k}.so"); shutil.copy(os.path.join(ROOT, "c-blosc_amd", "libblosc_amd.so"), dst)
    spec = importlib.util.spec_from_file_location(f"c_blosc_amd_{k}", os.path.join(ROOT, "c-blosc_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod); mod.LIB_PATH = dst; mod.load(); return mod
nchunks, csz = 128, 64 << 20
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
host = DATASETS["bench19"](csz); tmp = np.empty(csz + 16, np.uint8)
r = R.blosc_compress_ctx(5, 1, 8, csz, host.ctypes.data, tmp.ctypes.data, csz + 16, b"lz4", 0, 8)
dev = torch.device("cuda:0")
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev); back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
src = torch.from_numpy(host).to(dev).unsqueeze(0).expand(nchunks, csz).contiguous()
comp[:, :r].copy_(torch.from_numpy(tmp[:r].copy()).to(dev).unsqueeze(0).expand(nchunks, r))
stockcomp = comp.clone()
out = [f"buffers: comp {comp.data_ptr():#x} back {back.data_ptr():#x} src {src.data_ptr():#x}"]
mods, bds, bcs, arenas = [], [], [], []
for k in range(ncopies):
    m = load(k)
    bd = m.DeviceBatch([stockcomp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
    bc = m.DeviceBatch([src[i].data_ptr() for i in range(nchunks)], [csz] * nchunks, [comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks)
    assert bd.decompress() == 0 and bc.compress(8, 5, 1, b"lz4") == 0 and bd.decompress() == 0
    mods.append(m); bds.append(bd); bcs.append(bc); arenas.append(new_lines())
dec = [[] for _ in mods]; enc = [[] for _ in mods]
for rnd in range(4):
    for k, m in enumerate(mods):
        L = m.load(); L.blosc_gpu_profile(1); L.blosc_gpu_profile_reset()
        for _ in range(3): bds[k].decompress()
        for _ in range(2): bcs[k].compress(8, 5, 1, b"lz4")
        L.blosc_gpu_profile(0)
        d = m.profile_get("k_decode_streams"); e = m.profile_get("k_encode_streams")
        dec[k].append(d[0] / d[1]); enc[k].append(e[0] / e[1])
# phase 2: instance 0 (and the last one) give their arenas back and get new ones: does the speed belong to the instance or to its memory?
phase2 = []
for k in (0, ncopies - 1):
    L = mods[k].load(); L.blosc_init(); L.blosc_destroy(); new_lines()
    assert bds[k].decompress() == 0 and bds[k].decompress() == 0
    L.blosc_gpu_profile(1); L.blosc_gpu_profile_reset()
    for _ in range(4): bds[k].decompress()
    L.blosc_gpu_profile(0)
    d = mods[k].profile_get("k_decode_streams")
    phase2.append(f"instance {k} after giving its arenas back: decode {d[0] / d[1]:.3f} ms   arenas: " + " | ".join(a.split("arena", 1)[1].strip() for a in new_lines()))
# phase 3: instance 0 again after instance 1 has given up its arenas too (does it take over instance 1's memory?)
os.dup2(saved, 2)
for k in range(ncopies):
    addrs = " | ".join(a.split("arena", 1)[1].strip() for a in arenas[k])
    out.append(f"instance {k}: decode {np.median(dec[k]):.3f} ms (min {min(dec[k]):.3f})  encode {np.median(enc[k]):.3f} ms   arenas: {addrs}")
print("\n".join(out + phase2))
