#!/usr/bin/env python3
"""bench.py — the reference's headline benchmark (bench/bench.c "single" suite: compress then
decompress a 64 MiB chunk of the synthetic bench19 generator, report MB/s and ratio) scaled to the
MI355X configuration of BASELINE.json configs[1]:

    byte-shuffle + LZ4, clevel 5, typesize 8, 8 GiB synthetic (128 chunks x 64 MiB) on 1 x MI355X

One STEP = one compress pass + one decompress pass over the whole batch, all buffers resident in
HBM, through the C ABI of libblosc_amd (blosc_gpu_compress_batch / blosc_gpu_decompress_batch).
`value` = uncompressed bytes taken through that round trip per second, summed over all GPUs
(weak scaling: every rank owns its own 8 GiB; chunks are independent, no data-path collective).
Per-direction rates, per-kernel HIP-event times, the roofline of the dominant kernel, the
whole-direction roofline fractions ((nbytes + cbytes) / t / 8 TB/s) and the
reference's own multi-threaded CPU path on this box's host cores are reported in the same line.

Run:  python bench.py [--gpus N --steps K --warmup W]          (N>1: under torch.distributed.run)
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (6.29 TB/s measured copy)
KERNELS = ["k_shuffle", "k_bitshuffle", "k_encode_streams", "k_chunk_scan", "k_chunk_compact",
           "k_decode_plan", "k_decode_streams", "k_unshuffle", "k_bitunshuffle", "k_copy_chunks"]
COMPRESS_KERNELS = KERNELS[:5]
DECOMPRESS_KERNELS = KERNELS[5:]


def load_pkg():
    spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["c_blosc_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def make_chunk(kind, nbytes):
    from helpers import DATASETS
    return DATASETS[kind](nbytes)


def measured_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of THIS workload
    (profiles/r01_final_traffic.json, made by scripts/final_profile.sh + scripts/make_traffic_json.py:
    FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 corrections applied).  PMC counters cannot be
    read from inside a timed run, so other workloads than the default one report null."""
    default = (args.chunks == 128 and args.chunk_mib == 64 and args.typesize == 8 and args.clevel == 5 and
               args.shuffle == 1 and args.codec == "lz4" and args.data == "bench19")
    path = os.path.join(ROOT, "profiles", "r01_final_traffic.json")
    if not default or not os.path.exists(path):
        return None
    with open(path) as fh:
        k = json.load(fh)["kernels"].get(kernel)
    return k["hbm_bytes"] if k else None


def cpu_baseline(chunk_host, typesize, clevel, shuffle, cname, budget_passes):
    """The reference's own SSE2/AVX2 multi-threaded path (oracle/_ref/libblosc_ref.so, built from the
    reference sources) on this box's host cores; falls back to the single-threaded oracle port."""
    n = chunk_host.size
    refso = os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so")
    ncores = os.cpu_count() or 1
    out = {}
    if os.path.exists(refso):
        R = C.CDLL(refso)
        R.blosc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        R.blosc_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        R.blosc_init()
        nth = min(ncores, 256)
        R.blosc_set_nthreads(nth)
        R.blosc_set_compressor(cname)
        dests = [np.empty(n + 16, np.uint8) for _ in range(4)]     # bench.c uses several destination chunks
        back = np.empty(n, np.uint8)
        cb = R.blosc_compress(clevel, shuffle, typesize, n, chunk_host.ctypes.data, dests[0].ctypes.data, n + 16)  # warm-up
        R.blosc_decompress(dests[0].ctypes.data, back.ctypes.data, n)
        t0 = time.perf_counter()
        for i in range(budget_passes):
            cb = R.blosc_compress(clevel, shuffle, typesize, n, chunk_host.ctypes.data, dests[i % 4].ctypes.data, n + 16)
        t1 = time.perf_counter()
        for i in range(4):
            R.blosc_compress(clevel, shuffle, typesize, n, chunk_host.ctypes.data, dests[i].ctypes.data, n + 16)
        t2 = time.perf_counter()
        for i in range(budget_passes):
            R.blosc_decompress(dests[i % 4].ctypes.data, back.ctypes.data, n)
        t3 = time.perf_counter()
        assert np.array_equal(back, chunk_host)
        tc, td = (t1 - t0) / budget_passes, (t3 - t2) / budget_passes
        out = {"value": n / (tc + td) / 1e9, "unit": "GB/s", "cores": nth, "kind": "reference",
               "compress_GBps": n / tc / 1e9, "decompress_GBps": n / td / 1e9, "ratio": n / cb,
               "sample": f"{budget_passes} passes of one {n >> 20} MiB chunk each way through blosc_compress/"
                         f"blosc_decompress of the reference built from its own sources, nthreads={nth}"}
        R.blosc_destroy()
    else:
        from helpers import orc_compress, orc_decompress
        O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        O.orc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int]
        O.orc_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        passes = max(1, budget_passes // 32)
        t0 = time.perf_counter()
        for _ in range(passes):
            r, ch = orc_compress(O, chunk_host, typesize, clevel, shuffle, cname.decode())
        t1 = time.perf_counter()
        for _ in range(passes):
            orc_decompress(O, ch, n)
        t2 = time.perf_counter()
        tc, td = (t1 - t0) / passes, (t2 - t1) / passes
        out = {"value": n / (tc + td) / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
               "compress_GBps": n / tc / 1e9, "decompress_GBps": n / td / 1e9, "ratio": n / r,
               "sample": f"{passes} passes of one {n >> 20} MiB chunk through the scalar oracle port"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=128, help="chunks per GPU")
    ap.add_argument("--chunk-mib", type=int, default=64)
    ap.add_argument("--typesize", type=int, default=8)
    ap.add_argument("--clevel", type=int, default=5)
    ap.add_argument("--shuffle", type=int, default=1)
    ap.add_argument("--codec", default="lz4")
    ap.add_argument("--data", default="bench19")
    ap.add_argument("--cpu-passes", type=int, default=192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    mod = load_pkg()
    lib = mod.load()
    assert lib.blosc_gpu_set_device(local) == 0
    cname = args.codec.encode()

    nchunks, csz = args.chunks, args.chunk_mib << 20
    total = nchunks * csz
    host_chunk = make_chunk(args.data, csz)
    d_chunk = torch.from_numpy(host_chunk).to(dev)
    # distinct buffers per chunk (identical content, like bench.c's "i from 0 per chunk"): real HBM traffic
    src = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
    src.copy_(d_chunk.unsqueeze(0).expand(nchunks, csz))
    cstride = csz + 256
    comp = torch.empty((nchunks, cstride), dtype=torch.uint8, device=dev)
    back = torch.empty((nchunks, csz), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    bc = mod.DeviceBatch([src[i].data_ptr() for i in range(nchunks)], [csz] * nchunks,
                         [comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks)
    bd = mod.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks,
                         [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)

    def step():
        assert bc.compress(args.typesize, args.clevel, args.shuffle, cname, 0, stream) == 0
        assert bd.decompress(stream) == 0

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    cbytes = bc.results()
    assert all(c > 0 for c in cbytes), cbytes[:4]
    assert bd.results() == [csz] * nchunks, bd.results()[:4]

    lib.blosc_gpu_profile(1)
    lib.blosc_gpu_profile_reset()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    lib.blosc_gpu_profile(0)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    prof = {}
    for k in KERNELS:
        ms, cnt = mod.profile_get(k)
        if cnt:
            prof[k] = {"ms_avg": ms / cnt, "launches": cnt}
    sum_cb = float(sum(cbytes))
    t_c = sum(prof[k]["ms_avg"] for k in COMPRESS_KERNELS if k in prof) / 1e3
    t_d = sum(prof[k]["ms_avg"] for k in DECOMPRESS_KERNELS if k in prof) / 1e3

    # ---- verification (outside the timed region) ----
    verified = None
    if not args.no_verify:
        ok = bool(torch.equal(back, src))
        ch0 = comp[0][:cbytes[0]].cpu().numpy()
        refso = os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so")
        chk = np.empty(csz, np.uint8)
        if os.path.exists(refso):
            R = C.CDLL(refso)
            R.blosc_decompress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            r = R.blosc_decompress_ctx(ch0.ctypes.data, chk.ctypes.data, csz, 4)
            who = "stock c-blosc (oracle/_ref)"
        else:
            O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
            O.orc_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            r = O.orc_decompress(ch0.ctypes.data, chk.ctypes.data, csz)
            who = "oracle"
        ok = ok and r == csz and np.array_equal(chk, host_chunk)
        verified = {"roundtrip_bit_exact": ok, "gpu_chunk_decoded_by": who}
        assert ok, "verification failed"

    # ---- decompress of chunks written by the reference itself (drop-in direction) ----
    stock = None
    if rank == 0 or True:
        refso = os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so")
        ref_chunk = None
        if os.path.exists(refso):
            R = C.CDLL(refso)
            R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
            tmp = np.empty(csz + 16, np.uint8)
            r = R.blosc_compress_ctx(args.clevel, args.shuffle, args.typesize, csz, host_chunk.ctypes.data, tmp.ctypes.data, csz + 16, cname, 0, 1)
            if r > 0:
                ref_chunk = tmp[:r].copy()
        if ref_chunk is not None:
            comp[:, :ref_chunk.size].copy_(torch.from_numpy(ref_chunk).to(dev).unsqueeze(0).expand(nchunks, ref_chunk.size))
            back.zero_()
            assert bd.decompress(stream) == 0     # warm
            lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
            torch.cuda.synchronize(); ts = time.perf_counter()
            for _ in range(max(2, args.steps)):
                assert bd.decompress(stream) == 0
            torch.cuda.synchronize(); te = time.perf_counter()
            lib.blosc_gpu_profile(0)
            assert bd.results() == [csz] * nchunks
            assert torch.equal(back, src)
            tk = sum(mod.profile_get(k)[0] / max(mod.profile_get(k)[1], 1) for k in DECOMPRESS_KERNELS) / 1e3
            wall = (te - ts) / max(2, args.steps)
            stock = {"GBps_wall": total / wall / 1e9, "GBps_kernels": total / tk / 1e9, "ratio": csz / ref_chunk.size,
                     "roofline_frac": (total + nchunks * ref_chunk.size) / tk / 1e9 / HBM_PEAK_GBPS,
                     "k_decode_streams_ms": mod.profile_get("k_decode_streams")[0] / max(mod.profile_get("k_decode_streams")[1], 1),
                     "k_unshuffle_ms": mod.profile_get("k_unshuffle")[0] / max(mod.profile_get("k_unshuffle")[1], 1)}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel ----
    dom = max(prof, key=lambda k: prof[k]["ms_avg"])
    alg_bytes = total + sum_cb                      # nbytes + cbytes per chunk, x chunks per launch
    ach = alg_bytes / (prof[dom]["ms_avg"] / 1e3) / 1e9
    roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBPS, "traffic": measured_traffic(dom, args),
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": prof[dom]["ms_avg"]}
    out = {
        "metric": "compress+decompress GB/s (uncompressed) at 1/2/4/8 GPUs vs HBM roofline; ratio",
        "value": world * args.steps * total / elapsed / 1e9,
        "unit": "GB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"byte-shuffle + {args.codec} clevel={args.clevel} typesize={args.typesize}, "
                               f"{nchunks} x {args.chunk_mib} MiB {args.data} chunks per GPU ({total / 2**30:.0f} GiB), "
                               "step = compress pass + decompress pass, device-resident",
                   "codec": args.codec, "shuffle": args.shuffle, "typesize": args.typesize, "clevel": args.clevel,
                   "chunks_per_gpu": nchunks, "chunk_bytes": csz, "dataset": args.data},
        "ratio": total / sum_cb,
        "compress": {"GBps_kernels": total / t_c / 1e9, "roofline_frac_path": (total + sum_cb) / t_c / 1e9 / HBM_PEAK_GBPS},
        "decompress": {"GBps_kernels": total / t_d / 1e9, "roofline_frac_path": (total + sum_cb) / t_d / 1e9 / HBM_PEAK_GBPS},
        "decompress_stock_chunks": stock,
        "kernels": prof,
        "roofline": roof,
        "verified": verified,
    }
    if not args.no_cpu_baseline and world == 1:      # the host-core baseline is a 1-GPU artefact (rank 0, N = 1)
        out["cpu_baseline"] = cpu_baseline(host_chunk, args.typesize, args.clevel, args.shuffle, cname, args.cpu_passes)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
