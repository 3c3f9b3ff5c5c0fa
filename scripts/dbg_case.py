"""Decode one (data set, shuffle, typesize) case of reference-written chunks with every library named on the command line and say WHERE the output
differs from the plaintext (chunk, first offset, count, which byte planes).   python scripts/dbg_case.py lib.so ...   env: DATA SHUFFLE TYPESIZE CODEC CLEVEL CHUNKS"""
import ctypes as C, importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
def load(path, tag):
    spec = importlib.util.spec_from_file_location("c_blosc_amd_" + tag, os.path.join(ROOT, "c-blosc_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod); mod.LIB_PATH = os.path.abspath(path); mod.load(); return mod
libs = sys.argv[1:]
nchunks = int(os.environ.get("CHUNKS", "128")); csz = int(os.environ.get("CHUNK_MIB", "64")) << 20
dname, sh, ts = os.environ.get("DATA", "linspace"), int(os.environ.get("SHUFFLE", "1")), int(os.environ.get("TYPESIZE", "4"))
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
host = DATASETS[dname](csz); tmp = np.empty(csz + 16, np.uint8)
r = R.blosc_compress_ctx(int(os.environ.get("CLEVEL", "5")), sh, ts, csz, host.ctypes.data, tmp.ctypes.data, csz + 16, os.environ.get("CODEC", "lz4").encode(), 0, 8)
bs = int(tmp[8:12].view("<u4")[0]); print(f"{dname} shuffle={sh} T={ts}: cbytes {r} blocksize {bs} flags {tmp[2]:#x}")
dev = torch.device("cuda:0")
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev); back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
comp[:, :r].copy_(torch.from_numpy(tmp[:r].copy()).to(dev).unsqueeze(0).expand(nchunks, r))
want = torch.from_numpy(host).to(dev)
for k, p in enumerate(libs):
    m = load(p, str(k))
    b = m.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
    for rep in range(3):
        back.fill_(0xEE); rc = b.decompress(); res = b.results()
        bad = (back != want.unsqueeze(0))
        nbad = int(bad.sum())
        line = f"  {os.path.basename(p)} call {rep}: rc {rc} results ok {res == [csz] * nchunks} wrong bytes {nbad}"
        if nbad:
            ch = int(bad.any(dim=1).nonzero()[0]); row = bad[ch].nonzero().flatten()
            first = int(row[0]); last = int(row[-1])
            planes = sorted(set((row[:100000] % ts).tolist()))
            line += f"; bad chunks {int(bad.any(dim=1).sum())}, first chunk {ch}: offsets {first} .. {last} ({row.numel()} bytes), block {first // bs} .. {last // bs}, byte planes {planes}, got {back[ch][first:first+8].tolist()} want {want[first:first+8].tolist()}"
        print(line, flush=True)
