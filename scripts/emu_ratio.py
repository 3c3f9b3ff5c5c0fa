"""CPU, no GPU: compressed size of shuffled byte planes under the LZ4 encoders of csrc/enc_lz.h / enc_lz4p.h, run on the wavefront emulator
(tests/tools/liblz_wave_cpu.so - the same source the GPU runs).  What a change to the match finder costs in ratio is known before a GPU call:
    python scripts/emu_ratio.py [dataset ...]        KINDS=12,13,14  CLEVEL=5  T=8  BLOCKS=2  BSIZE=1048576
kinds: -1 = the reference's LZ4_compress_fast with the level's acceleration (oracle/_ref), 12 = the sequential select / emit loop, 13 = parallel parse probing every position, 14 = parallel parse probing every other position,
0 = what the library picks for the level, 1 = the BloscLZ writer (sequential finder).  Every stream is decoded again by the oracle's LZ4 decoder and compared with the plane."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS, ptr  # noqa: E402

E = C.CDLL(os.environ.get("EMU_LIB", os.path.join(ROOT, "tests", "tools", "liblz_wave_cpu.so")))
E.emu_lz_encode.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
O.orc_lz4_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
O.orc_blosclz_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
R = None
if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so")):
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so"))
    R.LZ4_compress_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]

kinds = [int(k) for k in os.environ.get("KINDS", "12,13,14").split(",")]
clevel = int(os.environ.get("CLEVEL", "5"))
T = int(os.environ.get("T", "8"))
nblocks = int(os.environ.get("BLOCKS", "2"))
bsize = int(os.environ.get("BSIZE", str(1 << 20)))
names = sys.argv[1:] or ["bench19", "linspace", "randwalk"]

for name in names:
    data = DATASETS[name](nblocks * bsize)
    line = f"{name:10s} T={T} cl{clevel}:"
    for kind in kinds:
        total = 0
        rv = C.c_ulonglong(0)
        rsum = 0
        t0 = time.time()
        for b in range(nblocks):
            blk = data[b * bsize:(b + 1) * bsize]
            planes = np.ascontiguousarray(blk.reshape(-1, T).T)
            for j in range(T):
                pl = planes[j]
                dst = np.zeros(pl.size + 64, np.uint8)
                if kind < 0:
                    r = R.LZ4_compress_fast(ptr(pl), ptr(dst), pl.size, pl.size, 10 - clevel)      # blosc/blosc.c:577-587
                else:
                    r = E.emu_lz_encode(kind, ptr(pl), pl.size, ptr(dst), pl.size, clevel, C.byref(rv))
                rsum += rv.value
                if r == 0:
                    total += pl.size
                    continue
                back = np.zeros(pl.size + 8, np.uint8)
                assert (O.orc_blosclz_decompress if kind == 1 else O.orc_lz4_decompress)(ptr(dst), r, ptr(back), pl.size) == pl.size, (name, kind, b, j)
                assert np.array_equal(back[:pl.size], pl), (name, kind, b, j)
                total += r
        line += f"  kind {kind}: ratio {nblocks * bsize / total:7.2f} ({rsum / 1e6:.2f} M rendezvous, {time.time() - t0:.0f} s)"
    print(line, flush=True)
