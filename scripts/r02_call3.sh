#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
python scripts/dbg_blocks.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== pytest subset"; timeout 900 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_spans.py tests/test_gpu_getitem_batch.py tests/test_gpu_compress.py -m gpu -q -x --no-header -p no:cacheprovider --timeout 300 2>&1 | tee gpurun_out/pytest_sub.log | tail -6
rm -f /tmp/bdprof.bin*; timeout 200 python scripts/bd_phase.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bd_phase_stock.log
for d in bench19 linspace randwalk; do DATA=$d timeout 100 python scripts/dec_sweep.py 2>&1 | grep -v amdgpu.ids; done
