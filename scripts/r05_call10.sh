#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05j
timeout 600 python -m pytest tests/test_gpu_random_streams.py tests/test_gpu_spans.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${T}_pytest.log | tail -6
echo "== decab"; ROUNDS=3 DECSETS="bench19:1:8 linspace:1:8 bench19:2:4 bench19:1:2" timeout 300 python scripts/dec_ab.py c-blosc_amd/libblosc_amd.so gpurun_tune_rowreg0.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_dec_ab.txt
CODEC=blosclz ROUNDS=3 DECSETS="bench19:1:8" timeout 200 python scripts/dec_ab.py c-blosc_amd/libblosc_amd.so gpurun_tune_rowreg0.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${T}_dec_ab.txt
