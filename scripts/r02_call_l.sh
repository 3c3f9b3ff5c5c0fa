#!/bin/bash
# Round 2, call l: the whole GPU suite on the current head + profile evidence for config 2 (kernel trace, FETCH / WRITE passes)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x 2>&1 | tee gpurun_out/l_pytest_gpu.log | tail -8
echo "== bench cfg 2"; timeout 300 python bench.py --config 2 2> gpurun_out/l_bench_cfg2.err | tee gpurun_out/l_bench_cfg2.json | cut -c1-300
CFG=2 TAG=r02f bash scripts/profile_config.sh 2>&1 | tail -30
for cfg in z 2c; do echo "== bench cfg $cfg"; timeout 300 python bench.py --config $cfg --no-cpu-baseline 2> gpurun_out/l_bench_cfg$cfg.err | tee gpurun_out/l_bench_cfg$cfg.json | cut -c1-300; done
