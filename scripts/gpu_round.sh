#!/bin/bash
# One gpurun call: quick sanity, GPU test-suite, bench.  Everything lands in gpurun_out/.
# Every stage has its own short timeout: a hung kernel must not eat the GPU budget.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== quick"; timeout 120 python tests/gpu_quick.py 2>&1 | tee gpurun_out/quick.log | tail -60
echo "== pytest gpu"; timeout ${PYTEST_TIMEOUT:-420} python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider --timeout 120 2>&1 | tee gpurun_out/pytest_gpu.log | tail -40
echo "== bench small"; timeout 150 python bench.py --chunks 16 --steps 3 --warmup 1 --cpu-passes 32 2>&1 | tee gpurun_out/bench_small.log | tail -5
echo "== bench full"; timeout 240 python bench.py 2>&1 | tee gpurun_out/bench_full.log | tail -5
