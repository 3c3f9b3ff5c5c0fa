"""GPU: the multi-GPU path of SURVEY §8e with the code that runs on a node - backend "nccl" (RCCL) - at the one
world size this box has: ONE logical chunk list, multigpu.chunk_range() for this rank, the batched call on that range,
gather_cbytes() over RCCL (communicator of size 1), and the checks a consumer of the consolidated table makes.
The world-size-2 logic (ragged ranges, padding of the all_gather) is covered on CPU with gloo in
tests/test_multigpu_gloo.py; bench.py --gpus N runs exactly this sequence per rank."""
import importlib.util
import os
import tempfile

import numpy as np
import pytest

from helpers import DATASETS, orc_decompress

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_batch_and_rccl_gather(pkg, oracle):
    import torch
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("c_blosc_amd_multigpu", os.path.join(ROOT, "c-blosc_amd", "multigpu.py"))
    multigpu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(multigpu)
    dev = torch.device("cuda:0")
    rdv = tempfile.NamedTemporaryFile(prefix="bamd_rdv_", delete=False)
    rdv.close()
    os.unlink(rdv.name)
    dist.init_process_group("nccl", init_method=f"file://{rdv.name}", rank=0, world_size=1, device_id=dev)
    try:
        nchunks, csz = 24, 4 << 20
        kinds = ["bench19", "linspace", "randwalk", "zeros"]
        host = [DATASETS[kinds[i % 4]](csz) for i in range(nchunks)]          # the logical chunk list
        lo, hi = multigpu.chunk_range(nchunks, dist.get_world_size(), dist.get_rank())
        assert (lo, hi) == (0, nchunks)
        src = torch.stack([torch.from_numpy(h) for h in host[lo:hi]]).to(dev)
        comp = torch.zeros((hi - lo, csz + 16), dtype=torch.uint8, device=dev)
        back = torch.zeros((hi - lo, csz), dtype=torch.uint8, device=dev)
        bc = pkg.DeviceBatch([src[i].data_ptr() for i in range(hi - lo)], [csz] * (hi - lo), [comp[i].data_ptr() for i in range(hi - lo)], [csz + 16] * (hi - lo))
        assert bc.compress(8, 5, 1, b"lz4") == 0
        cb = bc.results()
        assert all(c > 0 for c in cb)
        table, offsets = multigpu.gather_cbytes(cb, nchunks, device=dev)          # RCCL all_gather
        assert table == cb and offsets[0] == 0 and all(offsets[i + 1] == offsets[i] + cb[i] for i in range(nchunks - 1))
        assert all(multigpu.owner_of(c, nchunks, 1) == 0 for c in range(nchunks))
        # the consolidated view is enough to find and decode any chunk
        for i in (0, 7, nchunks - 1):
            r, out = orc_decompress(oracle, comp[i][:table[i]].cpu().numpy(), csz)
            assert r == csz and np.array_equal(out, host[i])
        bd = pkg.DeviceBatch([comp[i].data_ptr() for i in range(hi - lo)], table[lo:hi], [back[i].data_ptr() for i in range(hi - lo)], [csz] * (hi - lo))
        assert bd.decompress() == 0 and bd.results() == [csz] * (hi - lo)
        assert torch.equal(back, src)
        # payload consolidation over the same communicator: pack, all-gather-v, scatter back
        packed = multigpu.pack_local([comp[i] for i in range(hi - lo)], cb)
        container, offs = multigpu.gather_payload(packed, table, nchunks)
        assert offs == offsets and container.numel() == sum(cb)
        for i in (0, 11, nchunks - 1):
            assert torch.equal(container[offs[i]:offs[i] + cb[i]], comp[i][:cb[i]])
        owner, _ = multigpu.gather_payload(packed, table, nchunks, dst=0)
        mine, loff = multigpu.scatter_payload(owner, table, nchunks, src=0)
        assert torch.equal(mine, packed) and loff == offsets
        # byte counters / elapsed time are reduced the way bench.py does it
        t = torch.tensor([float(sum(cb))], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        assert float(t.item()) == float(sum(table))
    finally:
        dist.destroy_process_group()


def test_bench_spawner_n1():
    """bench.py's own launcher (the path `python bench.py --gpus N` takes for N > 1: re-execution under torch.distributed.run, one
    process per GPU, LOCAL_RANK -> blosc_gpu_set_device, RCCL communicator, one JSON line from rank 0) on the one GPU of this box."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "1", "--warmup", "1", "--chunks", "4",
                        "--chunk-mib", "8", "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["verified"]["roundtrip_bit_exact"] is True
    with open(os.path.join(ROOT, d["extra_file"])) as fh:      # everything that is not headline material
        x = json.load(fh)
    assert x["multi_gpu"]["rccl_world_size"] == 1 and len(x["multi_gpu"]["per_rank_GBps"]) == 1 and x["src_fingerprint"] == d["src_fingerprint"]


def test_bench_line_fits_the_drivers_tail():
    """The whole default protocol - headline workload, the five extra legs, the mixed batch, the stock-chunk legs, the CPU baseline - on small
    chunks: the ONE line must parse out of the last 4096 bytes of stdout (round 4's was 20.7 KB and the driver recorded `parsed: null`) and
    carry the fields the driver records; stderr must stay short as well (the driver's 8 KiB tail is stdout + stderr)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--chunks", "4", "--chunk-mib", "8", "--cpu-seconds", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(r.stderr) < 2048, r.stderr[-3000:]
    tail = r.stdout[-4096:]
    line = tail.strip().splitlines()[-1]
    assert len(line) < 4096 and line.startswith("{")
    d = json.loads(line)
    for k in ("value", "ms_per_step", "config", "roofline", "cpu_baseline", "legs", "extra_file"):
        assert k in d, k
    assert d["roofline"]["frac"] > 0 and d["roofline"]["decode_stock"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    assert set(d["legs"]) >= {"3", "4", "1g", "2t", "2x", "3e", "mixed"}
    assert os.path.exists(os.path.join(ROOT, d["extra_file"]))
