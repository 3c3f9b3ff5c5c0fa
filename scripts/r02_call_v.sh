#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_zlib.py tests/test_gpu_baseline_geometry.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x -k "zlib" 2>&1 | tail -5 | tee gpurun_out/v_tests.log
echo "== dec zlib"
for d in bench19 linspace randwalk; do
  CODEC=zlib CHUNKS=128 DATA=$d timeout 200 python scripts/dec_sweep.py 2>&1 | grep data=
done | tee gpurun_out/v_dec_zlib.log
