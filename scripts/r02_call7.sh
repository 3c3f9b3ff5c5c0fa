#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== pytest zstd"; timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_decompress.py tests/test_gpu_compress.py -m gpu -q -x --no-header -p no:cacheprovider --timeout 300 2>&1 | tail -5
for d in bench19 linspace randwalk; do CODEC=zstd CLEVEL=3 DATA=$d timeout 100 python scripts/dec_sweep.py 2>&1 | grep -v amdgpu.ids; done
