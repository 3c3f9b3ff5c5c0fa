#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05h
timeout 900 python -m pytest tests/test_gpu_parity_sweep.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${T}_pytest.log | tail -12
timeout 1200 python scripts/parity_hunt.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_parity_hunt.txt | tail -30
