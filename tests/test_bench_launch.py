"""`python bench.py --gpus N` must start its N ranks itself (`--gpus` was parsed and ignored).  CPU suite: the same
spawn path - bench.py re-executes itself under torch.distributed.run - on gloo with `--dry-gloo` (no GPU, no kernels): partition,
cbytes all_gather, payload all-gather-v, one JSON line from rank 0 with n_gpus = N.  The RCCL form of the same path at N = 1 is
tests/test_gpu_multigpu_nccl.py::test_bench_spawner_n1."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e)


@pytest.mark.parametrize("n", [1, 2, 3, 8])       # 8: the node size the scaling run uses
def test_dry_gloo_spawns_n_ranks(n):
    r = _run(["--gpus", str(n), "--dry-gloo", "--chunks", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                     # ONE line, from rank 0
    assert len(lines[0]) < 4096                          # the driver parses the line out of an 8 KiB tail of stdout + stderr (round 4: 20.7 KB, parsed: null)
    d = json.loads(r.stdout[-4096:].strip().splitlines()[-1])
    assert d["n_gpus"] == n and d["multi_gpu"]["world"] == n
    assert d["multi_gpu"]["chunks_per_rank"] == [5] * n and d["config"]["chunks_total"] == 5 * n
    assert d["metric"].startswith("compress+decompress GB/s") and d["scaling"] == "weak"


def test_gpus_flag_must_match_the_launcher():
    # under somebody else's launcher with a different world size: refuse, loudly
    r = _run(["--gpus", "4", "--dry-gloo"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_gpus_defaults_to_the_launchers_world_size():
    # `torchrun --nproc-per-node 2 bench.py --dry-gloo` without --gpus: one of the launcher's ranks, N = WORLD_SIZE
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--dry-gloo", "--chunks", "3"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["chunks_total"] == 6


def test_headline_line_assembly_stays_small():
    """assemble() on a record shaped like a full default run (five extra legs, mixed batch, ten CPU arms): the line must stay below the limit and
    carry the fields the driver records; everything else goes to the extra file."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    kern = {k: {"ms_avg": 1.2345678901234, "launches": 20} for k in bench.KERNELS}
    roof = {"bound": "hbm", "kernel": "k_encode_streams", "achieved": 1056.4458416437328, "peak": 8000.0, "unit": "GB/s", "frac": 0.1320557302054666,
            "traffic": None, "traffic_source": "profiles/r05_traffic_cfg2.json is stale (sources 0123456789ab, running ba9876543210)", "algorithmic_bytes_per_launch": 8766990336.0,
            "avg_launch_ms": 8.298570537567139, "path_frac": {"decompress": 0.31234567, "compress": 0.1312345678}, "per_rank_frac": [0.1320557302054666] * 8}
    roof["decode_stock"] = dict(roof, kernel="k_decode_streams", per_rank_frac=[0.29] * 8)
    roof["decode_stock"].pop("path_frac")
    stock = {"GBps_wall": 2200.123456, "GBps_kernels": 2300.123456, "ratio": 36.7029586005933, "roofline_frac": 0.29, "kernels_ms": {"k_decode_plan": 0.05123, "k_decode_streams": 3.8123456}, "roofline": roof["decode_stock"]}
    leg = {"value": 600.123456789, "unit": "GB/s", "steps": 5, "ms_per_step": 14.123456, "ratio": 113.7123456, "roofline": roof, "kernels": kern, "decompress_stock_chunks": stock, "verified": {"roundtrip_bit_exact": True}}
    res = {"value": 719.123456789, "unit": "GB/s", "steps": 20, "warmup": 5, "ms_per_step": 11.9123456789, "first_call_ms": 30.0, "sched_cold": {"ms_per_step": 13.0},
           "config": {"workload": "config #2: byte-shuffle + lz4 clevel=5 typesize=8, 128 x 64 MiB bench19 chunks per GPU (8 GiB), step = compress pass + decompress pass, device-resident",
                      "name": "2", "codec": "lz4", "shuffle": 1, "typesize": 8, "clevel": 5, "chunks_per_gpu": 128, "chunks_total": 1024, "chunk_bytes": 67108864, "dataset": "bench19", "direction": "compress+decompress"},
           "ratio": 47.8123456, "decompress": {}, "decompress_stock_chunks": stock, "multi_gpu": {"per_rank_GBps": [700.123456] * 8, "consolidation_ms": 0.5123}, "kernels": kern, "roofline": roof,
           "verified": {"roundtrip_bit_exact": True, "gpu_chunk_decoded_by": "stock c-blosc (oracle/_ref)"}, "compress": {}}
    arms = [{"mode": "one chunk, internal threads", "threads": t, "compress_GBps": 9.672699000741341, "decompress_GBps": 4.1, "roundtrip_GBps": 2.88, "passes": [200, 135]} for t in range(10)]
    cpu = {"value": 49.0123456, "unit": "GB/s", "cores": 256, "kind": "reference", "compress_GBps": 96.5, "decompress_GBps": 116.4, "ratio": 36.7, "best_arm": "256 chunks in parallel, 1 thread each (_ctx calls)",
           "nthreads1": {}, "arms": arms, "nproc": 256, "cpu_model": "AMD EPYC 9575F 64-Core Processor", "shuffle_accel": "Shuffle CPU Information:\nSSE2 available: True\n" * 8 + "Using AVX2 implementation",
           "sample": "64 MiB chunk(s) of the same data through blosc_compress_ctx/blosc_decompress_ctx of the reference built from its own sources; mean over >= 3 passes per arm, about 20 s in total"}
    class A: pass
    for world in (1, 8):
        line, detail = bench.assemble(res, world, cpu, {n: leg for n in ("3", "4", "1g", "2t", "2x")}, dict(leg, workload="mixed"), A())
        assert len(line) < 4096, len(line)
        d = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "extra_file"):
            assert k in d, k
        assert d["roofline"]["decode_stock"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and "arms" not in d["cpu_baseline"] and "kernels" not in d
        assert "arms" in detail["cpu_baseline"] and "extra_configs" in detail
    os.remove(os.path.join(ROOT, d["extra_file"]))
    try:
        os.remove(os.path.join(ROOT, "gpurun_out", "bench_extra.json"))
    except OSError:
        pass
