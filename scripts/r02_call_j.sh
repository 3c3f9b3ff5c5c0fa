#!/bin/bash
# Round 2, call j: Zlib decode on the GPU - tests and first timings
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_zlib.py tests/test_gpu_zstd.py tests/test_gpu_decompress.py -m gpu -q --no-header -p no:cacheprovider --timeout 300 -x 2>&1 | tail -15 | tee gpurun_out/j_tests.log
echo "== dec zlib"
for d in bench19 linspace randwalk; do
  CODEC=zlib CHUNKS=32 DATA=$d timeout 200 python scripts/dec_sweep.py 2>&1 | grep data=
done | tee gpurun_out/j_dec_zlib.log
