#!/usr/bin/env python3
"""Pack / unpack a file with the bloscpack-layout container of c-blosc_amd/blpk.py (chunks through the batched GPU calls).
usage: blpk_cli.py c <in> <out.blp> [--cname lz4 --clevel 5 --typesize 8 --chunk-mib 1 --shuffle 1 --checksum adler32]
       blpk_cli.py d <in.blp> <out>"""
import argparse, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
ap = argparse.ArgumentParser()
ap.add_argument("mode", choices=["c", "d"]); ap.add_argument("src"); ap.add_argument("dst")
ap.add_argument("--cname", default="lz4"); ap.add_argument("--clevel", type=int, default=5); ap.add_argument("--typesize", type=int, default=8)
ap.add_argument("--chunk-mib", type=float, default=1.0); ap.add_argument("--shuffle", type=int, default=1); ap.add_argument("--checksum", default="adler32")
a = ap.parse_args()
pkg = _load("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); blpk = _load("blpk", os.path.join(ROOT, "c-blosc_amd", "blpk.py"))
lib = pkg.load()
t0 = time.perf_counter()
if a.mode == "c":
    ck = {"none": 0, "adler32": 1, "crc32": 2}[a.checksum.lower()]
    n, b = blpk.pack_file(lib, a.src, a.dst, chunk_size=int(a.chunk_mib * (1 << 20)), typesize=a.typesize, clevel=a.clevel, shuffle=a.shuffle, cname=a.cname.encode(), checksum=ck)
    print(f"{n} chunks, {os.path.getsize(a.src)} -> {b} bytes in {time.perf_counter() - t0:.2f} s")
else:
    out = blpk.unpack_file(lib, a.src, a.dst)
    print(f"{out.size} bytes in {time.perf_counter() - t0:.2f} s")
