"""What the per-kernel hipEvent pairs (blosc_gpu_profile(1), which bench.py keeps switched on over its timed region because the roofline figures are
to be measured there) cost the step they measure: the same 20 steps of config 2 with the events on and off, taking turns.
    python scripts/prof_overhead.py           env: CHUNKS=128 ROUNDS=4 STEPS=20"""
import importlib.util, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
torch.cuda.init(); L = m.load()
nchunks = int(os.environ.get("CHUNKS", "128")); csz = 64 << 20
rounds = int(os.environ.get("ROUNDS", "4")); steps = int(os.environ.get("STEPS", "20"))
dev = torch.device("cuda:0")
src = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev)
out = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
src.copy_(torch.from_numpy(DATASETS["bench19"](csz)).to(dev).unsqueeze(0).expand(nchunks, csz))
bc = m.DeviceBatch([src[i].data_ptr() for i in range(nchunks)], [csz] * nchunks, [comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks)
for _ in range(3): assert bc.compress(8, 5, 1, b"lz4", 0) == 0
cb = bc.results()
bd = m.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], cb, [out[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
for _ in range(3): assert bd.decompress() == 0
res = {0: [], 1: []}
for r in range(rounds):
    for mode in (1, 0):
        L.blosc_gpu_profile(mode); L.blosc_gpu_profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            bc.compress(8, 5, 1, b"lz4", 0); bd.decompress()
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / steps * 1e3)
        L.blosc_gpu_profile(0)
print("ms per step, kernel events on :", " ".join(f"{v:.3f}" for v in res[1]), " median %.3f" % np.median(res[1]))
print("ms per step, kernel events off:", " ".join(f"{v:.3f}" for v in res[0]), " median %.3f" % np.median(res[0]))
