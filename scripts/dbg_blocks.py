"""Debug: where does the block decoder differ?  (host-pointer ABI, no torch)"""
import ctypes as C, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS, orc_compress, orc_decompress, header, ptr
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
pkg.load()
O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
sz, i, vp = C.c_size_t, C.c_int, C.c_void_p
O.orc_compress.argtypes = [i, i, sz, sz, vp, vp, sz, i, sz, i]
O.orc_decompress.argtypes = [vp, vp, sz]
nbad = 0
for n in [300001, 1 << 20]:
  for T in (4, 8):
    for dname in ["bench19", "randwalk", "zeros", "smallints", "linspace"]:
        data = DATASETS[dname](n)
        for clevel in (1, 5, 9):
            for who in ("oracle", "gpu"):
                if who == "oracle": r, ch = orc_compress(O, data, T, clevel, 1, "lz4")
                else: r, ch = pkg.compress(data, T, clevel, 1, b"lz4")
                r4, out = pkg.decompress(ch, n)
                if r4 != n or not np.array_equal(out, data):
                    nbad += 1
                    h = header(ch)
                    bad = np.nonzero(out[:n] != data)[0] if r4 == n else np.array([-1])
                    bs = h["blocksize"]
                    print(f"BAD n={n} T={T} {dname} cl={clevel} writer={who} r={r4} bs={bs} nbad={bad.size} first={bad[:6]} planes={np.unique(bad % T)[:8]} blocks={np.unique(bad // bs)[:6]} inblock_elem={(bad[:6] % bs) // T}", flush=True)
print("dbg done, bad =", nbad)
