"""Bug hunt on the device (not a test: a wide grid run once per round, scripts/r05_call8.sh): reference-written chunks of every data set x typesize x
filter x codec x clevel decoded here and compared byte for byte on the device; the same inputs compressed here and decoded by the reference.
Prints one line per failing cell and a summary.   env: CHUNK_MIB (default 16), NCH (2), CODECS (lz4,blosclz), CLEVELS (5,1,9), DATA"""
import ctypes as C, importlib.util, itertools, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
lib = mod.load()
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
R.blosc_decompress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
csz = int(os.environ.get("CHUNK_MIB", "16")) << 20; nch = int(os.environ.get("NCH", "2"))
dev = torch.device("cuda:0")
comp = torch.zeros((nch, csz + 256), dtype=torch.uint8, device=dev); back = torch.zeros((nch, csz), dtype=torch.uint8, device=dev)
tmp = np.empty(csz + 16, np.uint8); out = np.empty(csz, np.uint8)
TS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 64, 255]
codecs = os.environ.get("CODECS", "lz4,blosclz").split(",")
CLEVELS = tuple(int(x) for x in os.environ.get("CLEVELS", "5,1,9").split(","))
ncell = nbad = 0; t0 = time.time()
for dname in os.environ.get("DATA", "linspace,bench19,randwalk,smallints,arange").split(","):
    full = DATASETS[dname](csz)
    for odd in (0, 1):                                   # odd = 1: a chunk that ends in the middle of an element and of a block
        n = csz - (0 if not odd else 4099)
        data = full[:n]; d_data = torch.from_numpy(data).to(dev)
        src = d_data.unsqueeze(0).expand(nch, n).contiguous()
        for T, sh, codec, cl in itertools.product(TS, (1, 2, 0), codecs, CLEVELS):
            if (sh == 1 and T == 1) or (odd and cl != 5) or (sh == 0 and (T not in (1, 8) or cl != 5)): continue
            ncell += 1
            r = R.blosc_compress_ctx(cl, sh, T, n, data.ctypes.data, tmp.ctypes.data, n + 16, codec.encode(), 0, 16)
            assert r > 0
            comp[:, :r].copy_(torch.from_numpy(tmp[:r]).to(dev).unsqueeze(0).expand(nch, r))
            back.fill_(0xEE)
            bd = mod.DeviceBatch([comp[i].data_ptr() for i in range(nch)], [n + 16] * nch, [back[i].data_ptr() for i in range(nch)], [n] * nch)
            for rep in range(2):
                rc = bd.decompress()
                if rc != 0 or bd.results() != [n] * nch or not bool(torch.equal(back[:, :n], src)):
                    nbad += 1; print("BAD decode of reference-written chunks", dname, "n", n, "T", T, "shuffle", sh, codec, "clevel", cl, "call", rep, "results", bd.results()[:2], "wrong bytes", int((back[:, :n] != src).sum()), flush=True)
            bc = mod.DeviceBatch([src[i].data_ptr() for i in range(nch)], [n] * nch, [comp[i].data_ptr() for i in range(nch)], [n + 16] * nch)
            rc = bc.compress(T, cl, sh, codec.encode()); cb = bc.results()
            if rc != 0 or min(cb) <= 0:
                nbad += 1; print("BAD compress", dname, n, T, sh, codec, cl, cb, flush=True); continue
            ch = comp[nch - 1][:cb[-1]].cpu().numpy()
            rr = R.blosc_decompress_ctx(ch.ctypes.data, out.ctypes.data, n, 4)
            if rr != n or not np.array_equal(out[:n], data):
                nbad += 1; print("BAD the reference reading a chunk written here", dname, "n", n, "T", T, "shuffle", sh, codec, "clevel", cl, "ret", rr, flush=True)
            back.fill_(0xEE)
            bd2 = mod.DeviceBatch([comp[i].data_ptr() for i in range(nch)], [n + 16] * nch, [back[i].data_ptr() for i in range(nch)], [n] * nch)
            if bd2.decompress() != 0 or bd2.results() != [n] * nch or not bool(torch.equal(back[:, :n], src)):
                nbad += 1; print("BAD decode of own chunks", dname, "n", n, "T", T, "shuffle", sh, codec, "clevel", cl, flush=True)
    print(f"{dname}: {ncell} cells so far, {nbad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"parity hunt: {ncell} cells, {nbad} bad")
