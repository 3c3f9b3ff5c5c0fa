#!/bin/bash
# Round 2, final evidence: whole GPU suite, default bench line (with CPU baseline), kernel trace + FETCH / WRITE passes for configs 2, 3, 4,
# bench lines for the other configs.  Everything lands in gpurun_out/ (copied to profiles/r02g_* afterwards).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x 2>&1 | tee gpurun_out/g_pytest_gpu.log | tail -5
echo "== smoke"; timeout 120 python __graft_entry__.py smoke 2>&1 | tail -3
echo "== bench cfg 2"; timeout 300 python bench.py 2> gpurun_out/g_bench_cfg2.err | tee gpurun_out/g_bench_cfg2.json | cut -c1-200
for cfg in 2 3 4; do CFG=$cfg TAG=r02g bash scripts/profile_config.sh > gpurun_out/g_profile_cfg$cfg.log 2>&1; tail -3 gpurun_out/g_profile_cfg$cfg.log | cut -c1-160; done
for cfg in 2b 2c 2d 3 3b 3c 4 4b 4c 1g z zb; do echo "== bench cfg $cfg"; timeout 200 python bench.py --config $cfg --no-cpu-baseline 2> gpurun_out/g_bench_cfg$cfg.err | tee gpurun_out/g_bench_cfg$cfg.json | cut -c1-160; done
