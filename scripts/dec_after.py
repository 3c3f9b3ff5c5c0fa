"""What slows the decode kernel down inside bench.py's step (3.6 ms there, 3.3 ms on its own)?  The same decompress call timed
(a) back to back, (b) right after a compress call of the same batch (the bench's step), (c) right after 8 ms of matrix multiplies
(arithmetic, no memory traffic to speak of), (d) right after 16 GiB of device copies (memory traffic, no arithmetic).
    python scripts/dec_after.py            env: CHUNKS=128 ROUNDS=6"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
m = importlib.import_module("c-blosc_amd")
from helpers import DATASETS
nchunks = int(os.environ.get("CHUNKS", "128")); csz = 64 << 20; rounds = int(os.environ.get("ROUNDS", "6"))
dev = torch.device("cuda:0")
src = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
comp = torch.empty((nchunks, csz + 256), dtype=torch.uint8, device=dev)
back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
src.copy_(torch.from_numpy(DATASETS["bench19"](csz)).to(dev).unsqueeze(0).expand(nchunks, csz))
enc = m.DeviceBatch([src[i].data_ptr() for i in range(nchunks)], [csz] * nchunks, [comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks)
dec = m.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks, [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)
assert enc.compress(8, 5, 1, b"lz4", 0) == 0
a = torch.randn((8192, 8192), dtype=torch.bfloat16, device=dev); b = torch.randn((8192, 8192), dtype=torch.bfloat16, device=dev)
def mm():
    for _ in range(16): torch.matmul(a, b)
def cp():
    back.copy_(src); back.copy_(src)
L = m.load()
def timed_decode():
    L.blosc_gpu_profile(1); L.blosc_gpu_profile_reset(); dec.decompress(); L.blosc_gpu_profile(0)
    return m.profile_get("k_decode_streams")[0]
for _ in range(3): dec.decompress()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
for name, pre in (("back to back", None), ("after compress", lambda: enc.compress(8, 5, 1, b"lz4", 0)), ("after 16 matmuls", mm), ("after 16 GiB of copies", cp), ("back to back", None)):
    v = []; pv = []
    for _ in range(rounds):
        if pre:
            t0.record(); pre(); t1.record()
        r = timed_decode()
        if pre: torch.cuda.synchronize(); pv.append(t0.elapsed_time(t1))
        v.append(r)
    print(f"{name:24s} k_decode_streams {np.median(v):.3f} ms (min {min(v):.3f})" + (f"   [the work in front of it: {np.median(pv):.2f} ms]" if pv else ""), flush=True)
if os.environ.get("WITH_DIST"):          # the same once more with a process group like bench.py's (RCCL initialised, one barrier run)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dist.barrier(); torch.cuda.synchronize()
    v = [timed_decode() for _ in range(rounds)]
    print(f"with an RCCL process group    k_decode_streams {np.median(v):.3f} ms (min {min(v):.3f})", flush=True)
    v = []
    for _ in range(rounds):
        enc.compress(8, 5, 1, b"lz4", 0); v.append(timed_decode())
    print(f"... and after compress       k_decode_streams {np.median(v):.3f} ms (min {min(v):.3f})", flush=True)
