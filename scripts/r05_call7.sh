#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05g
timeout 1500 python -m pytest tests/test_gpu_spans.py tests/test_gpu_parity_sweep.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${T}_pytest.log | tail -25
