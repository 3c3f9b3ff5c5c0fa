#!/bin/bash
# Round 2, re-entry call: the whole GPU suite on the current head, the default bench line, per-dataset sweeps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== pytest gpu"; timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x --durations=8 2>&1 | tee gpurun_out/pytest_gpu.log | tail -25
echo "== bench cfg 2"; timeout 300 python bench.py --config 2 2> gpurun_out/bench_cfg2.err | tee gpurun_out/bench_cfg2.json | cut -c1-300
for cfg in 3 4 1g; do
  echo "== bench cfg $cfg"; timeout 200 python bench.py --config $cfg --no-cpu-baseline 2> gpurun_out/bench_cfg$cfg.err | tee gpurun_out/bench_cfg$cfg.json | cut -c1-300
done
echo "== enc sweep"; timeout 200 python scripts/enc_sweep.py 2>&1 | tee gpurun_out/enc_sweep.log
echo "== dec sweep"; for d in bench19 linspace zeros; do DATA=$d timeout 100 python scripts/dec_sweep.py 2>&1 | tee -a gpurun_out/dec_sweep.log; done
