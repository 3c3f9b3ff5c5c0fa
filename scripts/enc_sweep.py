"""Encode-only timing per dataset (device-resident batch), with a decode check."""
import ctypes as C, importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
lib = mod.load()
nchunks = int(os.environ.get("CHUNKS", "128")); csz = 64 << 20
CL = int(os.environ.get("CLEVEL", "5"))
dev = torch.device("cuda:0")
comp = torch.empty((nchunks, csz + 16), dtype=torch.uint8, device=dev)
back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
src = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
for dname in os.environ.get("DATA", "bench19,linspace,randwalk,random,zeros").split(","):
    host = DATASETS[dname](csz)
    src.copy_(torch.from_numpy(host).to(dev).unsqueeze(0).expand(nchunks, csz))
    for codec in os.environ.get("CODECS", "lz4").split(","):
        bc = mod.DeviceBatch([src[i].data_ptr() for i in range(nchunks)], [csz] * nchunks, [comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks)
        bc.compress(8, CL, 1, codec.encode(), 0)
        lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
        for _ in range(2): bc.compress(8, CL, 1, codec.encode(), 0)
        lib.blosc_gpu_profile(0)
        cb = bc.results()
        e = mod.profile_get("k_zstd_encode" if codec == "zstd" else ("k_zlib_encode" if codec == "zlib" else "k_encode_streams")); s = mod.profile_get("k_shuffle")
        bd = mod.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], cb, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
        bd.decompress()
        ok = bool((back == src).all())
        print(f"data={dname:9s} codec={codec:8s} ratio={csz/cb[0]:8.2f}: k_encode_streams {e[0]/e[1]:8.3f} ms  k_shuffle {s[0]/max(s[1],1):6.3f} ms  ok={ok}", flush=True)
