#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import ctypes as C, importlib.util, os, sys, time
import numpy as np
ROOT = os.getcwd(); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS, orc_decompress, ref_decompress, header, ptr
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
pkg.load()
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
R.blosc_decompress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
for dname in ["bench19", "linspace", "randwalk", "zeros", "random", "smallints"]:
    for n in [1000, 300001, 4 << 20]:
        for T, sh, cl in [(8, 1, 3), (4, 2, 5), (8, 1, 9), (1, 0, 1)]:
            data = DATASETS[dname](n)
            r, ch = pkg.compress(data, T, cl, sh, b"zstd")
            ok = False; r3 = None
            if r > 0:
                r3, out3 = ref_decompress(R, ch, n)
                ok = r3 == n and np.array_equal(out3, data)
                r4, out4 = pkg.decompress(ch, n)
                ok = ok and r4 == n and np.array_equal(out4, data)
            tmp = np.zeros(n + 16, np.uint8)
            rr = R.blosc_compress_ctx(cl, sh, T, n, ptr(data), ptr(tmp), n + 16, b"zstd", 0, 4)
            print(f"{dname:9s} n={n:8d} T={T} sh={sh} cl={cl}: gpu {r:9d} (ratio {n/max(r,1):8.2f})  stock ratio {n/max(rr,1):8.2f}  stock-reads-it={ok} {r3}", flush=True)
PY
