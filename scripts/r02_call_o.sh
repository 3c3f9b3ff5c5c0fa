#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_modes.py tests/test_gpu_baseline_geometry.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x 2>&1 | tail -4 | tee gpurun_out/o_tests.log
for cfg in 4 4b 4c; do echo "== bench cfg $cfg"; timeout 300 python bench.py --config $cfg --no-cpu-baseline 2> gpurun_out/o_bench_cfg$cfg.err | tee gpurun_out/o_bench_cfg$cfg.json | cut -c1-200; done
