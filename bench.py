#!/usr/bin/env python3
"""bench.py — the reference's headline benchmark (bench/bench.c "single" suite: compress then
decompress 64 MiB chunks of a synthetic generator, report MB/s and ratio; protocol bench.c:250-320)
scaled to the MI355X configurations of BASELINE.json / SURVEY.md §8d:

    --config 2   byte-shuffle + LZ4  clevel 5, typesize 8, bench19   (default; BASELINE.json configs[1])
    --config 2b / 2c / 2d            same call on linspace / random-walk float64 / incompressible bytes
    --config 3   bitshuffle + LZ4    clevel 5, typesize 4, bench19   (3b: arange, 3c: small random ints)
    --config 4   byte-shuffle + Zstd clevel 3, typesize 8, bench19   (4b/4c/4d: the other float64 sets)
    --config 1g  byte-shuffle + BloscLZ clevel 5, typesize 8, bench19 (config #1's call, on the GPU)
    --config z   byte-shuffle + Zlib clevel 5, typesize 8, bench19   (the remaining codec; zb: linspace)

every one as 128 chunks x 64 MiB = 8 GiB per GPU (512 chunks = 32 GiB per GPU when N > 1: config #5).

One STEP = one compress pass + one decompress pass over the rank's chunks, all buffers resident in
HBM, through the C ABI of libblosc_amd (blosc_gpu_compress_batch / blosc_gpu_decompress_batch).
`value` = uncompressed bytes taken through that round trip per second, summed over all GPUs.
Multi-GPU: ONE logical list of `chunks_per_gpu x N` chunks is partitioned into contiguous
ranges (c-blosc_amd/multigpu.py: chunk_range); chunks are independent, so there is no data-path
collective - the only exchange is the all_gather of the per-chunk cbytes table over RCCL
(gather_cbytes, backend "nccl", also with a communicator of size 1), timed separately as
`consolidation_ms`.  The JSON line also carries per-direction rates, per-kernel HIP-event times, the
roofline of the dominant kernel, decompression of chunks written by the reference itself (the drop-in
direction) and the reference's own multi-threaded CPU path on this box's host cores.

Run:  python bench.py [--config C --gpus N --steps K --warmup W]
      N > 1: `python bench.py --gpus N` starts the N ranks itself (it re-executes itself under torch.distributed.run, one process
      per GPU, the way the reference's entry point starts its own workers - blosc/blosc.c:1890-1949 init_threads); launched under
      torch.distributed.run by somebody else (RANK / WORLD_SIZE in the environment) it is one of those ranks.
      `--dry-gloo` runs the same launch + partition + exchanges + JSON assembly on the CPU over gloo (no GPU, no kernels; CPU suite).
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (6.29 TB/s measured copy)
COMPRESS_KERNELS = ["k_shuffle", "k_bitshuffle", "k_encode_streams", "k_lz4hc_encode", "k_zstd_encode", "k_zlib_encode", "k_chunk_scan", "k_chunk_compact"]
DECOMPRESS_KERNELS = ["k_decode_plan", "k_decode_streams", "k_zstd_entropy", "k_zstd_seq", "k_zstd_exec", "k_zstd_streams", "k_zlib_streams", "k_unshuffle",
                      "k_bitunshuffle", "k_copy_chunks"]
KERNELS = COMPRESS_KERNELS + DECOMPRESS_KERNELS
REFSO = os.path.join(ROOT, "oracle", "_ref", "libblosc_ref.so")

# SURVEY.md §8d "Concrete inputs"
CONFIGS = {
    "2":  dict(codec="lz4", shuffle=1, typesize=8, clevel=5, data="bench19"),
    "2b": dict(codec="lz4", shuffle=1, typesize=8, clevel=5, data="linspace"),
    "2c": dict(codec="lz4", shuffle=1, typesize=8, clevel=5, data="randwalk"),
    "2d": dict(codec="lz4", shuffle=1, typesize=8, clevel=5, data="random"),
    "3":  dict(codec="lz4", shuffle=2, typesize=4, clevel=5, data="bench19"),
    "3b": dict(codec="lz4", shuffle=2, typesize=4, clevel=5, data="arange"),
    "3c": dict(codec="lz4", shuffle=2, typesize=4, clevel=5, data="smallints"),
    "3e": dict(codec="lz4", shuffle=2, typesize=8, clevel=5, data="bench19"),      # bitshuffle at typesize 8 (float64 + bitshuffle: inside the codec kernels since round 5)
    "4":  dict(codec="zstd", shuffle=1, typesize=8, clevel=3, data="bench19"),
    "4b": dict(codec="zstd", shuffle=1, typesize=8, clevel=3, data="linspace"),
    "4c": dict(codec="zstd", shuffle=1, typesize=8, clevel=3, data="randwalk"),
    "4d": dict(codec="zstd", shuffle=1, typesize=8, clevel=3, data="random"),
    "1g": dict(codec="blosclz", shuffle=1, typesize=8, clevel=5, data="bench19"),
    # typesize 2 and 16 (round 3: their byte (un)shuffle runs inside the codec kernels like that of 4 and 8)
    "2t": dict(codec="lz4", shuffle=1, typesize=2, clevel=5, data="bench19"),
    "2x": dict(codec="lz4", shuffle=1, typesize=16, clevel=5, data="bench19"),
    "z":  dict(codec="zlib", shuffle=1, typesize=8, clevel=5, data="bench19"),
    "zb": dict(codec="zlib", shuffle=1, typesize=8, clevel=5, data="linspace"),
    # the encoder options of DESIGN.md 3.6 / 3.9 (an "env" entry is put into the environment before the library is used)
    "h":  dict(codec="lz4hc", shuffle=1, typesize=8, clevel=9, data="bench19"),
    "4p": dict(codec="zstd", shuffle=1, typesize=8, clevel=3, data="bench19", env={"BLOSC_AMD_ZSTD_TABLES": "0"}),    # predefined tables (the round-2 default)
    "4t": dict(codec="zstd", shuffle=1, typesize=8, clevel=3, data="bench19", env={"BLOSC_AMD_ZSTD_TABLES": "1"}),
    "zf": dict(codec="zlib", shuffle=1, typesize=8, clevel=5, data="bench19", env={"BLOSC_AMD_ZLIB_DYNAMIC": "0", "BLOSC_AMD_ZLIB_SEARCH": "0"}),   # fixed codes, plain match finder (the round-2 default)
    "zy": dict(codec="zlib", shuffle=1, typesize=8, clevel=5, data="bench19", env={"BLOSC_AMD_ZLIB_DYNAMIC": "1", "BLOSC_AMD_ZLIB_SEARCH": "0"}),   # dynamic codes only
    "4s": dict(codec="zstd", shuffle=1, typesize=8, clevel=3, data="bench19", env={"BLOSC_AMD_ZSTD_SEARCH": "1"}),
    "zs": dict(codec="zlib", shuffle=1, typesize=8, clevel=5, data="bench19", env={"BLOSC_AMD_ZLIB_SEARCH": "1", "BLOSC_AMD_ZLIB_DYNAMIC": "0"}),
    "zd": dict(codec="zlib", shuffle=1, typesize=8, clevel=5, data="bench19", env={"BLOSC_AMD_ZLIB_DYNAMIC": "1", "BLOSC_AMD_ZLIB_SEARCH": "1"}),
    "4h": dict(codec="zstd", shuffle=1, typesize=4, clevel=3, data="smallints", env={"BLOSC_AMD_ZSTD_TABLES": "1", "BLOSC_AMD_ZSTD_HUFFMAN": "1"}),
    "4r": dict(codec="zstd", shuffle=1, typesize=4, clevel=3, data="smallints"),       # 4h's baseline: raw literals, predefined tables
}
FILTER_NAME = {0: "no filter", 1: "byte-shuffle", 2: "bitshuffle"}


def load_pkg():
    spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["c_blosc_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def make_chunk(kind, nbytes):
    from helpers import DATASETS
    return DATASETS[kind](nbytes)


def _strip_comments(text):
    """C / C++ source without comments and with runs of white space collapsed (string and character literals are kept as they are)"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c); i += 1
    return " ".join("".join(out).split())


def src_fingerprint():
    """12 hex digits over the CODE of every file of c-blosc_amd/csrc (the kernels and the engine; comments and white space do not count): stamps
    the traffic files and the extra file, so that a figure measured on other kernels than the ones running is never quoted (.git does not travel to
    the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "c-blosc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        with open(os.path.join(d, f), "r", errors="replace") as fh:
            h.update(_strip_comments(fh.read()).encode())
    return h.hexdigest()[:12]


def measured_traffic(kernel, config_name, nchunks, chunk_mib, stock=False):
    """(HBM bytes per launch of `kernel`, where they come from) out of the committed rocprofv3 PMC passes of THIS workload
    (profiles/r<NN>_traffic_cfg<config>.json, made by scripts/profile_config.sh + scripts/make_traffic_json.py: FETCH_SIZE and WRITE_SIZE
    in separate passes, gfx950 corrections applied).  PMC counters cannot be read from inside a timed run, so the figure is the committed
    one - and only when the file was made from the SAME kernel sources as the running build (`src_fingerprint` inside it); a file of
    other sources, or of another geometry than 128 x 64 MiB, gives null and says why.  stock=True: the launches that decoded
    REFERENCE-written chunks ("kernels_stock")."""
    if nchunks != 128 or chunk_mib != 64:
        return None, "no PMC pass of this geometry"
    # the newest round's file whose sources are the running ones (r06_traffic_cfg2.json, r05_...: one per round that ran the passes)
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_traffic_cfg{config_name}.json")), reverse=True)
    if not cands:
        return None, "no PMC pass of this configuration on the round's kernels"
    fp, stale = src_fingerprint(), None
    for path in cands:
        with open(path) as fh:
            doc = json.load(fh)
        rel = os.path.relpath(path, ROOT)
        if doc.get("src_fingerprint") != fp:
            stale = stale or f"{rel} is stale (sources {doc.get('src_fingerprint')}, running {fp})"
            continue
        k = (doc.get("kernels_stock") or {}).get(kernel) if stock else None
        k = k or doc["kernels"].get(kernel)
        return (k["hbm_bytes"] if k else None), rel
    return None, stale

# ---------------------------------------------------------------------------------------------
# CPU baseline: the reference itself (oracle/_ref, built from /root/reference's sources) on this box's cores
# ---------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _capture_stdout(fn):
    """Runs fn() with file descriptor 1 redirected to a temp file (the reference prints its shuffle
    implementation with printf when BLOSC_PRINT_SHUFFLE_ACCEL is set, shuffle.c:258-273)."""
    libc = C.CDLL(None)
    sys.stdout.flush()
    saved = os.dup(1)
    with tempfile.TemporaryFile() as tf:
        os.dup2(tf.fileno(), 1)
        try:
            fn()
            libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        tf.seek(0)
        return tf.read().decode(errors="replace").strip()


def _shuffle_accel_of_reference():
    """what BLOSC_PRINT_SHUFFLE_ACCEL=1 makes the reference print on its first shuffle (shuffle.c:258-273), from a process of its own"""
    import subprocess
    code = ("import ctypes as C, numpy as np\n"
            f"R = C.CDLL({REFSO!r})\n"
            "R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]\n"
            "a = np.arange(1 << 16, dtype=np.uint8); d = np.empty(a.size + 16, np.uint8)\n"
            "R.blosc_compress_ctx(5, 1, 8, a.size, a.ctypes.data, d.ctypes.data, d.size, b'lz4', 0, 1)\n"
            "C.CDLL(None).fflush(None)\n")
    try:
        env = dict(os.environ, BLOSC_PRINT_SHUFFLE_ACCEL="1")
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120)
        return r.stdout.decode(errors="replace").strip()
    except Exception as e:      # the baseline is context, never a reason to lose the line
        return f"(not captured: {e})"


def _mem_available():
    try:
        with open("/proc/meminfo") as fh:
            for ln in fh:
                if ln.startswith("MemAvailable"):
                    return int(ln.split()[1]) * 1024
    except OSError:
        pass
    return 64 << 30


def host_abi_rate(lib, chunk_host, typesize, clevel, shuffle, cname, reps=5):
    """What an UNMODIFIED c-blosc caller gets: blosc_compress / blosc_decompress of the stock ABI on host buffers (pageable numpy memory), one caller,
    one chunk at a time, PCIe both ways included (the number next to `cpu_baseline`).  Never `value`: the headline is device-resident."""
    import ctypes as C
    n = chunk_host.size
    src = np.ascontiguousarray(chunk_host); dst = np.empty(n + 16, np.uint8); back = np.empty(n, np.uint8)
    lib.blosc_init()
    try:
        lib.blosc_set_compressor(cname)
        cb = lib.blosc_compress(clevel, shuffle, typesize, n, src.ctypes.data, dst.ctypes.data, n + 16)
        assert cb > 0, cb
        assert lib.blosc_decompress(dst.ctypes.data, back.ctypes.data, n) == n
        t0 = time.perf_counter()
        for _ in range(reps):
            cb = lib.blosc_compress(clevel, shuffle, typesize, n, src.ctypes.data, dst.ctypes.data, n + 16)
        t1 = time.perf_counter()
        for _ in range(reps):
            r = lib.blosc_decompress(dst.ctypes.data, back.ctypes.data, n)
        t2 = time.perf_counter()
        assert r == n and np.array_equal(back, src)
    finally:
        lib.blosc_set_compressor(b"blosclz")
        lib.blosc_destroy()
    c, d = n * reps / (t1 - t0) / 1e9, n * reps / (t2 - t1) / 1e9
    return {"compress_GBps": c, "decompress_GBps": d, "round_trip_GBps": n * reps / (t2 - t0) / 1e9, "unit": "GB/s",
            "sample": f"{reps} x one {n >> 20} MiB chunk through blosc_compress / blosc_decompress on pageable host memory, one caller, PCIe both ways included"}


def cpu_baseline(chunk_host, typesize, clevel, shuffle, cname, budget_s=20.0):
    """SURVEY §8d / BASELINE.md §4: nthreads = 1, a sweep up to nproc (cap 256, blosc.h:51) and — because one
    64 MiB chunk has only 64-128 blocks for the pool to share — `P` independent blosc_compress_ctx /
    blosc_decompress_ctx calls running in parallel on P chunks (the chunk-parallel use the GPU batch mirrors).
    `value` is the best round-trip rate found; every arm is listed."""
    n = chunk_host.size
    ncores = os.cpu_count() or 1
    info = {"nproc": ncores, "cpu_model": _cpu_model()}
    if not os.path.exists(REFSO):
        from helpers import orc_compress, orc_decompress
        O = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        O.orc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int]
        O.orc_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        t0 = time.perf_counter()
        r, ch = orc_compress(O, chunk_host, typesize, clevel, shuffle, cname.decode())
        t1 = time.perf_counter()
        orc_decompress(O, ch, n)
        t2 = time.perf_counter()
        tc, td = t1 - t0, t2 - t1
        return {"value": n / (tc + td) / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                "compress_GBps": n / tc / 1e9, "decompress_GBps": n / td / 1e9, "ratio": n / r, **info,
                "sample": f"one pass of one {n >> 20} MiB chunk through the scalar oracle port"}
    # shuffle.c:258-273 prints the chosen implementation once, from the pthread_once that the FIRST call into the library
    # triggers - and this process has already used the reference (stock chunks).  Ask a fresh process.
    info["shuffle_accel"] = _shuffle_accel_of_reference()
    R = C.CDLL(REFSO)
    sz, i, vp = C.c_size_t, C.c_int, C.c_void_p
    R.blosc_compress_ctx.argtypes = [i, i, sz, sz, vp, vp, sz, C.c_char_p, sz, i]
    R.blosc_decompress_ctx.argtypes = [vp, vp, sz, i]
    dests = [np.empty(n + 16, np.uint8) for _ in range(4)]     # bench.c uses several destination chunks
    back = np.empty(n, np.uint8)
    cb = [0]

    def one_c(k, nth):
        cb[0] = R.blosc_compress_ctx(clevel, shuffle, typesize, n, chunk_host.ctypes.data, dests[k % 4].ctypes.data, n + 16, cname, 0, nth)

    def one_d(k, nth):
        r = R.blosc_decompress_ctx(dests[k % 4].ctypes.data, back.ctypes.data, n, nth)
        assert r == n

    for k in range(4):
        one_c(k, 1)
    one_d(0, 1)
    assert np.array_equal(back, chunk_host)
    ratio = n / cb[0]
    arms = []
    sweep = sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t <= ncores} | {min(ncores, 256)})
    per_arm = budget_s / (2 * (len(sweep) + 1))

    def timed(fn, nth):
        fn(0, nth)
        cnt, t0 = 0, time.perf_counter()
        best = 1e9
        while True:
            ta = time.perf_counter()
            fn(cnt, nth)
            best = min(best, time.perf_counter() - ta)
            cnt += 1
            if cnt >= 3 and (time.perf_counter() - t0 > per_arm or cnt >= 200):
                break
        return best, (time.perf_counter() - t0) / cnt, cnt

    for nth in sweep:
        bc_, mc, kc = timed(one_c, nth)
        bd_, md, kd = timed(one_d, nth)
        arms.append({"mode": "one chunk, internal threads", "threads": nth, "compress_GBps": n / mc / 1e9, "decompress_GBps": n / md / 1e9,
                     "roundtrip_GBps": n / (mc + md) / 1e9, "compress_best_GBps": n / bc_ / 1e9, "decompress_best_GBps": n / bd_ / 1e9,
                     "passes": [kc, kd]})
    # chunk-parallel arm: P threads, one chunk each, nthreads = 1 inside (ctypes releases the GIL)
    # every hardware thread gets a chunk (cap: blosc's own 256, blosc.h:51, and a third of the free memory)
    P = max(1, min(ncores, 256, int(_mem_available() // (3 * (n + 16))) or 1))
    pd = [np.empty(n + 16, np.uint8) for _ in range(P)]
    pb = [np.empty(n, np.uint8) for _ in range(P)]
    reps = 4

    def par(fn):
        ths = [threading.Thread(target=fn, args=(k,)) for k in range(P)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return time.perf_counter() - t0

    def pc_(k):
        for _ in range(reps):
            R.blosc_compress_ctx(clevel, shuffle, typesize, n, chunk_host.ctypes.data, pd[k].ctypes.data, n + 16, cname, 0, 1)

    def pd_(k):
        for _ in range(reps):
            R.blosc_decompress_ctx(pd[k].ctypes.data, pb[k].ctypes.data, n, 1)

    par(pc_)
    tc = par(pc_) / reps
    td = par(pd_) / reps
    assert np.array_equal(pb[P - 1], chunk_host)
    arms.append({"mode": f"{P} chunks in parallel, 1 thread each (_ctx calls)", "threads": P, "compress_GBps": P * n / tc / 1e9,
                 "decompress_GBps": P * n / td / 1e9, "roundtrip_GBps": P * n / (tc + td) / 1e9, "passes": [reps, reps]})
    best = max(arms, key=lambda a: a["roundtrip_GBps"])
    one = arms[0]
    return {"value": best["roundtrip_GBps"], "unit": "GB/s", "cores": best["threads"], "kind": "reference",
            "compress_GBps": best["compress_GBps"], "decompress_GBps": best["decompress_GBps"], "ratio": ratio,
            "best_arm": best["mode"], "nthreads1": {k: one[k] for k in ("compress_GBps", "decompress_GBps", "roundtrip_GBps")},
            "arms": arms, **info,
            "sample": f"{n >> 20} MiB chunk(s) of the same data through blosc_compress_ctx/blosc_decompress_ctx of the reference built "
                      f"from its own sources; mean over >= 3 passes per arm, about {budget_s:.0f} s in total"}


class Rig:
    """everything one process sets up once: torch / RCCL, the library, the rank's share of the chunk list"""
    pass


def measure(rig, name, cfg, steps, warmup, args, mixed=None, payload=False):
    """One workload through the timed protocol of the contract: `warmup` untimed steps, then exactly `steps` steps between
    barrier + synchronize pairs, MAX over ranks.  Returns the fields of the JSON line that belong to this workload."""
    torch, dist, mod, lib, multigpu, dev = rig.torch, rig.dist, rig.mod, rig.lib, rig.multigpu, rig.dev
    world, rank = rig.world, rig.rank
    cname = cfg["codec"].encode()
    T, clevel, shuffle = cfg["typesize"], cfg["clevel"], cfg["shuffle"]
    gpu_can_encode = lib.blosc_compname_to_compcode(cname) >= 0
    nchunks, csz, total = rig.nchunks, rig.csz, rig.nchunks * rig.csz
    src, comp, back = rig.src, rig.comp, rig.back
    datasets = mixed or [cfg["data"]]
    hosts = [make_chunk(d, csz) for d in datasets]
    for k, h in enumerate(hosts):                 # chunk i holds dataset i mod len(datasets): distinct buffers, real HBM traffic
        d_chunk = torch.from_numpy(h).to(dev)
        src[k::len(hosts)].copy_(d_chunk.unsqueeze(0).expand(len(range(k, nchunks, len(hosts))), csz))
    host_chunk = hosts[0]
    stream = torch.cuda.current_stream().cuda_stream
    bc = mod.DeviceBatch([src[i].data_ptr() for i in range(nchunks)], [csz] * nchunks,
                         [comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks)
    bd = mod.DeviceBatch([comp[i].data_ptr() for i in range(nchunks)], [csz + 16] * nchunks,
                         [back[i].data_ptr() for i in range(nchunks)], [csz] * nchunks)

    # the reference's own chunk of this data (drop-in direction; the only source of chunks for a codec the GPU cannot encode)
    ref_chunk = None
    if os.path.exists(REFSO) and not mixed and not (args.no_stock and gpu_can_encode):
        R = C.CDLL(REFSO)
        R.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
        tmp = np.empty(csz + 16, np.uint8)
        r = R.blosc_compress_ctx(clevel, shuffle, T, csz, host_chunk.ctypes.data, tmp.ctypes.data, csz + 16, cname, 0, min(os.cpu_count() or 1, 16))
        if r > 0:
            ref_chunk = tmp[:r].copy()

    def load_stock_chunks():
        comp[:, :ref_chunk.size].copy_(torch.from_numpy(ref_chunk).to(dev).unsqueeze(0).expand(nchunks, ref_chunk.size))

    decode_only = not gpu_can_encode
    if decode_only:
        assert ref_chunk is not None, f"the GPU cannot encode '{cfg['codec']}' and oracle/_ref is not built: nothing to decode"
        load_stock_chunks()

    def step():
        if not decode_only:
            assert bc.compress(T, clevel, shuffle, cname, 0, stream) == 0
        assert bd.decompress(stream) == 0

    def sync_all():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    first_ms = None
    for w in range(warmup):
        if w == 0:
            sync_all(); tf = time.perf_counter()
        step()
        if w == 0:
            torch.cuda.synchronize(); first_ms = (time.perf_counter() - tf) * 1e3
    cbytes = [ref_chunk.size] * nchunks if decode_only else bc.results()
    assert all(c > 0 for c in cbytes), cbytes[:4]
    assert bd.results() == [csz] * nchunks, bd.results()[:4]

    lib.blosc_gpu_profile(1)
    lib.blosc_gpu_profile_reset()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    lib.blosc_gpu_profile(0)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    per_rank_t = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(per_rank_t, t)
    per_rank_gbps = [steps * total / float(x.item()) / 1e9 for x in per_rank_t]
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- one step WITHOUT the queue order learnt from the previous calls (blosc_gpu_profile(2): plain block order, what a first call on new data gets) ----
    lib.blosc_gpu_profile(2)
    sync_all(); tc = time.perf_counter()
    step()
    torch.cuda.synchronize()
    sched_cold_ms = (time.perf_counter() - tc) * 1e3
    lib.blosc_gpu_profile(0)
    step()                                                   # (the next call's queues are built from this one's costs again)

    # ---- consolidation: every rank learns the global cbytes table (RCCL all_gather, 4 bytes per chunk) ----
    torch.cuda.synchronize()
    tg = time.perf_counter()
    table, offsets = multigpu.gather_cbytes(cbytes, rig.nchunks_total, device=dev)
    torch.cuda.synchronize()
    consolidation_ms = (time.perf_counter() - tg) * 1e3
    assert len(table) == rig.nchunks_total and table[rig.lo:rig.hi] == list(cbytes)
    sum_cb_global = float(sum(table))
    # ---- and the payloads: this rank's chunks packed back to back, then the all-gather-v onto rank 0, timed apart ----
    payload_ms = None
    if payload:
        torch.cuda.synchronize()
        tp = time.perf_counter()
        packed = multigpu.pack_local([comp[i] for i in range(nchunks)], list(cbytes))
        container, _ = multigpu.gather_payload(packed, table, rig.nchunks_total, dst=0)
        torch.cuda.synchronize()
        payload_ms = (time.perf_counter() - tp) * 1e3
        assert container is None or container.numel() == int(sum_cb_global)
        del packed, container

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
        who = None
        if not decode_only:
            ch0 = comp[0][:cbytes[0]].cpu().numpy()
            chk = np.empty(csz, np.uint8)
            if os.path.exists(REFSO):
                R = C.CDLL(REFSO)
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
    if ref_chunk is not None and not args.no_stock:
        if not decode_only:
            load_stock_chunks()
        back.zero_()
        assert bd.decompress(stream) == 0     # warm
        lib.blosc_gpu_profile(1); lib.blosc_gpu_profile_reset()
        torch.cuda.synchronize(); ts = time.perf_counter()
        reps = max(2, steps)
        for _ in range(reps):
            assert bd.decompress(stream) == 0
        torch.cuda.synchronize(); te = time.perf_counter()
        lib.blosc_gpu_profile(0)
        assert bd.results() == [csz] * nchunks
        assert torch.equal(back, src)
        kms = {k: mod.profile_get(k)[0] / max(mod.profile_get(k)[1], 1) for k in DECOMPRESS_KERNELS if mod.profile_get(k)[1]}
        tk = sum(kms.values()) / 1e3
        wall = (te - ts) / reps
        kdom = max(kms, key=kms.get)
        s_traffic, s_src = (None, None) if (rig.overridden or mixed) else measured_traffic(kdom, name, nchunks, args.chunk_mib, stock=True)
        stock = {"GBps_wall": total / wall / 1e9, "GBps_kernels": total / tk / 1e9, "ratio": csz / ref_chunk.size,
                 "roofline_frac": (total + nchunks * ref_chunk.size) / tk / 1e9 / HBM_PEAK_GBPS, "kernels_ms": kms,
                 "roofline": {"bound": "hbm", "kernel": kdom, "avg_launch_ms": kms[kdom], "algorithmic_bytes_per_launch": float(total + nchunks * ref_chunk.size),
                              "achieved": (total + nchunks * ref_chunk.size) / (kms[kdom] / 1e3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": (total + nchunks * ref_chunk.size) / (kms[kdom] / 1e3) / 1e9 / HBM_PEAK_GBPS,
                              "path_frac": (total + nchunks * ref_chunk.size) / tk / 1e9 / HBM_PEAK_GBPS,      # over ALL kernels of the direction
                              "traffic": s_traffic, "traffic_source": s_src}}

    # ---- roofline of the dominant kernel ----
    dom = max(prof, key=lambda k: prof[k]["ms_avg"])
    filt_only = dom in ("k_shuffle", "k_unshuffle", "k_bitshuffle", "k_bitunshuffle")
    # nbytes + cbytes per chunk for a codec (or fused) kernel, 2 x nbytes for a filter-only kernel
    alg_bytes = 2.0 * total if filt_only else total + sum_cb
    ach = alg_bytes / (prof[dom]["ms_avg"] / 1e3) / 1e9
    r_traffic, r_src = (None, None) if (rig.overridden or mixed) else measured_traffic(dom, name, nchunks, args.chunk_mib)
    roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBPS, "traffic": r_traffic, "traffic_source": r_src,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": prof[dom]["ms_avg"],
            # every pass of a direction counted (a three-pass pipeline cannot hide behind its best kernel): algorithmic bytes / sum of the direction's kernels
            "path_frac": {"decompress": (total + sum_cb) / t_d / 1e9 / HBM_PEAK_GBPS}}
    if not decode_only:
        roof["path_frac"]["compress"] = (total + sum_cb) / t_c / 1e9 / HBM_PEAK_GBPS
    if stock is not None:      # the north-star direction inside the object the driver records: decode of reference-written chunks
        roof["decode_stock"] = stock["roofline"]
    # every rank's own figure (weak scaling: each rank runs the same kernels on its own chunks), gathered to all
    mine = torch.tensor([roof["frac"], stock["roofline"]["frac"] if stock is not None else 0.0], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    roof["per_rank_frac"] = [round(float(x[0].item()), 4) for x in allr]
    if stock is not None:
        roof["decode_stock"]["per_rank_frac"] = [round(float(x[1].item()), 4) for x in allr]
    what = "decompress pass of reference-written chunks (the GPU does not encode this codec)" if decode_only else "compress pass + decompress pass"
    res = {
        "value": world * steps * total / elapsed / 1e9, "unit": "GB/s", "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "first_call_ms": first_ms,             # the first step of this workload in this process: cold queues (no cost feedback yet), arenas still growing
        "sched_cold": {"ms_per_step": sched_cold_ms, "GBps": world * total / (sched_cold_ms / 1e3) / 1e9,      # one step in plain block order, arenas warm
                       "note": "all 128 chunks of the bench are the same 64 MiB (bench/bench.c:383) and the queue order is trained on them; this is the step without that"},
        "config": {"workload": f"config #{name}: {FILTER_NAME[shuffle]} + {cfg['codec']} clevel={clevel} typesize={T}, "
                               f"{nchunks} x {args.chunk_mib} MiB {'+'.join(datasets)} chunks per GPU ({total / 2**30:.0f} GiB), "
                               f"step = {what}, device-resident",
                   "name": name, "codec": cfg["codec"], "shuffle": shuffle, "typesize": T, "clevel": clevel,
                   "chunks_per_gpu": nchunks, "chunks_total": rig.nchunks_total, "chunk_bytes": csz, "dataset": "+".join(datasets),
                   "direction": "decompress" if decode_only else "compress+decompress"},
        "ratio": rig.nchunks_total * csz / sum_cb_global,
        "decompress": {"GBps_kernels": total / t_d / 1e9, "roofline_frac_path": (total + sum_cb) / t_d / 1e9 / HBM_PEAK_GBPS},
        "decompress_stock_chunks": stock,
        "multi_gpu": {"partition": "contiguous chunk ranges, no data-path collective", "consolidation": "all_gather of the cbytes table, backend nccl (RCCL)",
                      "consolidation_ms": consolidation_ms, "world": world, "rccl_world_size": dist.get_world_size(), "per_rank_GBps": per_rank_gbps,
                      "payload_consolidation": "pack + all-gather-v of the compressed chunks onto rank 0 (grouped send/recv, counts from the table)",
                      "payload_consolidation_ms": payload_ms, "payload_bytes": sum_cb_global,
                      "note": None if world > 1 else "communicator of size 1: no N > 1 line exists until the driver has a multi-GPU node"},
        "kernels": prof,
        "roofline": roof,
        "verified": verified,
    }
    if not decode_only:
        res["compress"] = {"GBps_kernels": total / t_c / 1e9, "roofline_frac_path": (total + sum_cb) / t_c / 1e9 / HBM_PEAK_GBPS}
    res["_host_chunk"] = host_chunk
    return res


LINE_LIMIT = 4000      # bytes; the driver keeps an 8 KiB tail of stdout + stderr together and parses the line out of it (round 4: 20.7 KB -> parsed: null)


def _r(x, nd=4):
    """floats of the headline line carry what they mean, not 17 digits"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}") if abs(x) < 1 else round(x, 3)
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, list):
        return [_r(v, nd) for v in x]
    return x


def assemble(res, world, cpu, extra, mixed, args):
    """(the ONE line for stdout, the full record).  The line holds the contract's fields, `config`, `roofline` (with `decode_stock`, the
    north-star kernel on reference-written chunks), `cpu_baseline` without its arms, `verified`, a five-number summary per extra leg and
    the name of the file with everything else (kernel tables, arms of the CPU baseline, legs in full, the mixed batch, multi-GPU detail):
    profiles/bench_extra_<fingerprint of the kernel sources>.json and gpurun_out/bench_extra.json (the copy gpurun brings back)."""
    fp = src_fingerprint()
    detail = dict(res)
    detail["src_fingerprint"] = fp
    detail["argv"] = sys.argv[1:]
    if extra is not None:
        detail["extra_configs"], detail["mixed_batch"] = extra, mixed
    if cpu is not None:
        detail["cpu_baseline"] = cpu
    extra_file = None
    for d in ("profiles", "gpurun_out"):
        try:
            os.makedirs(os.path.join(ROOT, d), exist_ok=True)
            name = f"bench_extra_{fp}.json" if d == "profiles" else "bench_extra.json"
            with open(os.path.join(ROOT, d, name), "w") as fh:
                json.dump(detail, fh, indent=1)
            extra_file = extra_file or f"{d}/{name}"
        except OSError:
            pass
    roof = dict(res["roofline"])
    stock = res.get("decompress_stock_chunks")
    k = res["kernels"]
    out = {
        "metric": "compress+decompress GB/s (uncompressed) at 1/2/4/8 GPUs vs HBM roofline; ratio",
        "value": res["value"], "unit": res["unit"], "n_gpus": world, "steps": res["steps"], "warmup": res["warmup"],
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {kk: res["config"][kk] for kk in ("workload", "name", "chunks_per_gpu", "chunks_total", "chunk_bytes")},
        "ratio": res["ratio"],
        "compress_ms": sum(k[n]["ms_avg"] for n in COMPRESS_KERNELS if n in k) or None,
        "decompress_ms": sum(k[n]["ms_avg"] for n in DECOMPRESS_KERNELS if n in k),
        "decompress_stock_ms": sum(stock["kernels_ms"].values()) if stock else None,
        "ratio_stock": stock["ratio"] if stock else None,
        "sched_cold_ms": res["sched_cold"]["ms_per_step"] if "sched_cold" in res else None,      # one step in plain block order (no cost feedback from an earlier call), arenas warm
        "roofline": roof,
        "verified": res["verified"],
    }
    if world > 1:
        out["per_rank_GBps"] = res["multi_gpu"]["per_rank_GBps"]
        out["consolidation_ms"] = res["multi_gpu"]["consolidation_ms"]
    if cpu is not None:
        out["cpu_baseline"] = {kk: cpu[kk] for kk in ("value", "unit", "cores", "kind", "compress_GBps", "decompress_GBps", "ratio", "best_arm", "nproc", "cpu_model") if kk in cpu}
        acc = cpu.get("shuffle_accel", "")
        out["cpu_baseline"]["shuffle_accel"] = "avx2" if "Using AVX2" in acc or "avx2" in acc.lower().split("using")[-1] else acc[-60:]
        out["cpu_baseline"]["sample"] = cpu["sample"]
    if extra is not None:      # [round-trip GB/s, ratio, compress ms, decompress ms of own chunks, decompress ms of reference-written chunks]
        def five(r):
            kk = r["kernels"]; st = r.get("decompress_stock_chunks")
            return [r["value"], r["ratio"], sum(kk[n]["ms_avg"] for n in COMPRESS_KERNELS if n in kk), sum(kk[n]["ms_avg"] for n in DECOMPRESS_KERNELS if n in kk),
                    sum(st["kernels_ms"].values()) if st else None]
        out["legs"] = {"columns": "GBps, ratio, compress_ms, decompress_ms, decompress_stock_ms", **{n: five(r) for n, r in extra.items()}}
        if mixed is not None:
            out["legs"]["mixed"] = [mixed["value"], mixed["ratio"], None, None, None]
    if "host_abi" in res:       # the stock ABI on host buffers, one caller, one 64 MiB chunk, PCIe included: [compress, decompress, round trip] GB/s - to be read next to cpu_baseline
        out["host_abi_GBps"] = [res["host_abi"]["compress_GBps"], res["host_abi"]["decompress_GBps"], res["host_abi"]["round_trip_GBps"]]
    out["extra_file"] = extra_file
    out["src_fingerprint"] = fp
    out = _r(out)
    line = json.dumps(out)
    for drop in ("legs", ("cpu_baseline", "sample"), ("roofline", "per_rank_frac")):      # never needed so far: the line is about 2.5 KB
        if len(line) <= LINE_LIMIT:
            break
        if isinstance(drop, tuple):
            out.get(drop[0], {}).pop(drop[1], None)
        else:
            out.pop(drop, None)
        line = json.dumps(out)
    assert len(line) <= LINE_LIMIT, len(line)
    return line, detail


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: re-execute this file under torch.distributed.run, one process per GPU
    (LOCAL_RANK -> blosc_gpu_set_device), on 127.0.0.1 and a free port.  The children inherit stdout: rank 0 prints the one JSON line.
    (The reference's entry points start their own workers the same way: blosc/blosc.c:1890-1949 init_threads, :871-899 parallel_blosc.)"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def _load_multigpu():
    mspec = importlib.util.spec_from_file_location("c_blosc_amd_multigpu", os.path.join(ROOT, "c-blosc_amd", "multigpu.py"))
    m = importlib.util.module_from_spec(mspec)
    mspec.loader.exec_module(m)
    return m


def dry_gloo(args):
    """The N-rank protocol without a GPU (CPU suite): every rank takes its range of ONE logical chunk list, "compresses" it (a cbytes
    figure per chunk from a fixed rule - no kernels exist here, and nothing is timed as if they did), and the exchanges and the JSON
    assembly of the real run follow on gloo: all_gather of the cbytes table, all-gather-v of the payloads onto rank 0, MAX over ranks
    of the elapsed time, per-rank figures gathered to rank 0, one line from rank 0."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    sys.stdout.flush()
    saved_fd1 = os.dup(1)        # gloo reports its connections through C stdio: stdout is for the one JSON line (see main)
    os.dup2(2, 1)
    try:
        if world > 1:
            dist.init_process_group("gloo")
        else:
            rdv = tempfile.NamedTemporaryFile(prefix="bamd_rdv_", delete=False); rdv.close(); os.unlink(rdv.name)
            dist.init_process_group("gloo", init_method=f"file://{rdv.name}", rank=0, world_size=1)
        dist.barrier()
        C.CDLL(None).fflush(None)
    finally:
        os.dup2(saved_fd1, 1)
        os.close(saved_fd1)
    mg = _load_multigpu()
    per_gpu = args.chunks or 8
    csz = 4096
    total_chunks = per_gpu * world
    lo, hi = mg.chunk_range(total_chunks, world, rank)
    rule = lambda c: 64 + (c * 37) % 191                          # the stand-in for a compressed size
    dist.barrier(); t0 = time.perf_counter()
    cbytes = [rule(c) for c in range(lo, hi)]
    rows = [torch.full((csz,), c % 251, dtype=torch.uint8) for c in range(lo, hi)]
    dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per_rank = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(per_rank, torch.tensor([elapsed, float(hi - lo)], dtype=torch.float64))
    table, offsets = mg.gather_cbytes(cbytes, total_chunks)
    assert table == [rule(c) for c in range(total_chunks)], "cbytes table differs from the rule"
    packed = mg.pack_local(rows, cbytes)
    container, offs = mg.gather_payload(packed, table, total_chunks, dst=0)
    if rank == 0:
        assert container.numel() == sum(table)
        for c in (0, total_chunks // 2, total_chunks - 1):       # chunk c's bytes are where the table says
            assert int(container[offs[c]]) == c % 251 and int(container[offs[c] + table[c] - 1]) == c % 251
        print(json.dumps({
            "metric": "compress+decompress GB/s (uncompressed) at 1/2/4/8 GPUs vs HBM roofline; ratio",
            "value": None, "unit": "GB/s", "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic", "dry": True,
            "config": {"workload": f"dry run on gloo: {per_gpu} stand-in chunks per rank, no kernels", "chunks_per_gpu": per_gpu, "chunks_total": total_chunks},
            "multi_gpu": {"world": world, "backend": "gloo", "chunks_per_rank": [int(p[1].item()) for p in per_rank], "payload_bytes": int(sum(table)),
                          "partition": "contiguous chunk ranges, no data-path collective"},
        }))
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node (default: WORLD_SIZE when a launcher started this process, else 1)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS))
    ap.add_argument("--chunks", type=int, default=0, help="chunks per GPU (default 128 = 8 GiB at EVERY N, so that the N = 1 point of a scaling run is the headline run; 512 = config #5's 32 GiB share)")
    ap.add_argument("--chunk-mib", type=int, default=64)
    ap.add_argument("--typesize", type=int, default=None)
    ap.add_argument("--clevel", type=int, default=None)
    ap.add_argument("--shuffle", type=int, default=None)
    ap.add_argument("--codec", default=None)
    ap.add_argument("--data", default=None)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-stock", action="store_true", help="skip the decompression of reference-written chunks")
    ap.add_argument("--no-extra", action="store_true", help="default config only: skip the short legs for configs #3 / #4 and the mixed batch")
    ap.add_argument("--spawn", action="store_true", help="start the ranks through torch.distributed.run also when N = 1 (the N > 1 launch path on a one-GPU box)")
    ap.add_argument("--dry-gloo", action="store_true", help="no GPU: launch, partition, cbytes / payload exchanges and the JSON line over gloo on the CPU (tests)")
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ["WORLD_SIZE"]) if "RANK" in os.environ and "WORLD_SIZE" in os.environ else 1
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    # ---- N > 1 and nobody has started the ranks: start them (one process per GPU), rank 0's JSON line is this process's output ----
    if (args.gpus > 1 or args.spawn) and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch with --nproc-per-node {args.gpus} (or let bench.py start the ranks itself)")
    if args.dry_gloo:
        return dry_gloo(args)
    cfg = dict(CONFIGS[args.config])
    os.environ.update(cfg.pop("env", {}))
    for k in ("typesize", "clevel", "shuffle", "codec", "data"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    overridden = any(getattr(args, k) is not None for k in ("typesize", "clevel", "shuffle", "codec", "data"))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() < world:
        sys.exit(f"bench.py: --gpus {world} but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    # RCCL communicator also at N = 1 (size 1): the code path is the same for every N.
    # RCCL prints a version banner through C stdio when the communicator comes up; stdout is for the one JSON line, so
    # fd 1 points at stderr until the first collective is through and C stdio is flushed.
    libc = C.CDLL(None)
    sys.stdout.flush()
    saved_fd1 = os.dup(1)
    os.dup2(2, 1)
    try:
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)
        else:
            rdv = tempfile.NamedTemporaryFile(prefix="bamd_rdv_", delete=False)
            rdv.close()
            os.unlink(rdv.name)
            dist.init_process_group("nccl", init_method=f"file://{rdv.name}", rank=0, world_size=1, device_id=dev)
        dist.barrier()
        torch.cuda.synchronize()
        libc.fflush(None)
    finally:
        os.dup2(saved_fd1, 1)
        os.close(saved_fd1)
    rig = Rig()
    rig.torch, rig.dist, rig.dev, rig.world, rig.rank, rig.overridden = torch, dist, dev, world, rank, overridden
    rig.mod = load_pkg()
    rig.lib = rig.mod.load()
    rig.multigpu = _load_multigpu()
    assert rig.lib.blosc_gpu_set_device(local) == 0

    per_gpu = args.chunks or 128
    rig.nchunks_total = per_gpu * world
    rig.lo, rig.hi = rig.multigpu.chunk_range(rig.nchunks_total, world, rank)
    rig.nchunks, rig.csz = rig.hi - rig.lo, args.chunk_mib << 20
    rig.src = torch.empty((rig.nchunks, rig.csz), dtype=torch.uint8, device=dev)
    rig.comp = torch.empty((rig.nchunks, rig.csz + 256), dtype=torch.uint8, device=dev)
    rig.back = torch.empty((rig.nchunks, rig.csz), dtype=torch.uint8, device=dev)

    res = measure(rig, args.config, cfg, args.steps, args.warmup, args, payload=True)
    host_chunk = res.pop("_host_chunk")
    # ---- the other BASELINE.json configurations and a batch of mixed data, as short legs of the default run:
    #      each with its own roofline object (dominant kernel, algorithmic bytes, average launch) and its ratio next to stock's ----
    extra = None
    mixed = None
    if args.config == "2" and not overridden and not args.no_extra and world == 1:
        extra = {}
        for name in ("3", "4", "1g", "2t", "2x", "3e", "2b", "2c", "2d", "3b", "3c"):       # 2b / 2c / 2d, 3b / 3c: the other data classes of configs 2 and 3 (linspace, random walk, random bytes -> MEMCPYED; arange, small integers - bench/bench.c:141-170 is one generator of several); 1g: config #1's call (BloscLZ) on the GPU; 2t / 2x: config 2 at typesize 2 and 16, 3e: config 3 at typesize 8 (the reference's bench takes the typesize as an argument over the same data, bench/bench.c:250-320)
            r = measure(rig, name, dict(CONFIGS[name]), min(args.steps, 5), 1, args)
            r.pop("_host_chunk")
            keep = ("value", "unit", "steps", "ms_per_step", "first_call_ms", "sched_cold", "ratio", "roofline", "kernels", "decompress_stock_chunks", "verified", "compress", "decompress")
            extra[name] = {k: r[k] for k in keep if k in r}
            extra[name]["workload"] = r["config"]["workload"]
        m = measure(rig, "2", cfg, min(args.steps, 5), 1, args, mixed=["bench19", "linspace", "randwalk", "random"])
        m.pop("_host_chunk")
        mixed = {k: m[k] for k in ("value", "unit", "steps", "ms_per_step", "first_call_ms", "ratio", "kernels", "verified")}
        mixed["workload"] = m["config"]["workload"]
    if rank != 0:
        dist.destroy_process_group()
        return
    cpu = None
    if not args.no_cpu_baseline and world == 1:      # the host-core baseline is a 1-GPU artefact (rank 0, N = 1)
        cpu = cpu_baseline(host_chunk, cfg["typesize"], cfg["clevel"], cfg["shuffle"], cfg["codec"].encode(), args.cpu_seconds)
    if extra is not None:
        res["host_abi"] = host_abi_rate(rig.lib, host_chunk, cfg["typesize"], cfg["clevel"], cfg["shuffle"], cfg["codec"].encode())
    line, detail = assemble(res, world, cpu, extra, mixed, args)
    print(line)
    sys.stdout.flush()
    rig.lib.blosc_init(); rig.lib.blosc_destroy()          # releases the arenas (and prints the BLOSC_AMD_HOSTTIME summary when that switch is on)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
