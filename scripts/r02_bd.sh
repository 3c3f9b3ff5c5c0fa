#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp BLOSC_AMD_BLOCKDEC=1
make -C oracle oracle > /dev/null 2>&1
python scripts/dbg_blocks.py 2>&1 | grep -v amdgpu.ids | tail -6
timeout 600 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_spans.py tests/test_gpu_getitem_batch.py tests/test_gpu_compress.py -m gpu -q -x --no-header -p no:cacheprovider --timeout 300 2>&1 | tail -3
for d in bench19 linspace randwalk; do DATA=$d python scripts/dec_sweep.py 2>&1 | grep -v amdgpu; done
rm -f /tmp/bdprof.bin*; timeout 200 python scripts/bd_phase.py 2>&1 | grep -v amdgpu.ids
