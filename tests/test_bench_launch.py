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
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["multi_gpu"]["world"] == n
    assert d["multi_gpu"]["chunks_per_rank"] == [5] * n and d["config"]["chunks_total"] == 5 * n
    assert d["metric"].startswith("compress+decompress GB/s") and d["scaling"] == "weak"


def test_gpus_flag_must_match_the_launcher():
    # under somebody else's launcher with a different world size: refuse, loudly
    r = _run(["--gpus", "4", "--dry-gloo"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
