#!/bin/bash
# round 5, final pass on the round's last kernel sources: whole GPU suite, default bench line, rocprofv3 kernel stats + FETCH / WRITE passes of
# configs 2 and 3, FETCH / WRITE of the decode of reference-written chunks.  Then, locally: scripts/make_traffic_json.py r05 2 ; ... r05 3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05_final4
timeout 1500 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${T}_pytest_gpu.log | tail -6
echo "== default bench"; timeout 400 python bench.py --steps 20 --warmup 5 2> gpurun_out/${T}_bench_default.err > gpurun_out/${T}_bench_default.json; wc -c gpurun_out/${T}_bench_default.json gpurun_out/${T}_bench_default.err; cut -c1-700 gpurun_out/${T}_bench_default.json
cp gpurun_out/bench_extra.json gpurun_out/${T}_bench_extra.json 2>/dev/null
for cfg in 2 3; do echo "== profile cfg $cfg"; CFG=$cfg TAG=r05 timeout 700 bash scripts/profile_config.sh 2>&1 | tail -25; done
echo "== stock decode traffic"; TAG=r05 timeout 400 bash scripts/gpu_call.sh dectraffic 2>&1 | tail -8
