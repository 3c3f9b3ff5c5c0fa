#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
timeout 600 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_spans.py tests/test_gpu_getitem_batch.py -m gpu -q -x --no-header -p no:cacheprovider --timeout 300 2>&1 | tail -3
for d in bench19 linspace randwalk; do DATA=$d python scripts/dec_sweep.py 2>&1 | grep -v amdgpu; done
WRITER=gpu python scripts/dec_phase.py 2>&1 | grep -E "kernel|plane 1|plane 2"
python scripts/dec_phase.py 2>&1 | grep -E "kernel|plane 1|plane 2"
