#!/bin/bash
# round 5, call 5: unshuffle group depth on data whose planes are real loads (raw splits, noisy planes)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05e
ROUNDS=3 DECSETS="randwalk:1:8 random:1:8 randwalk:1:4 linspace:1:4 randwalk:1:16" timeout 600 python scripts/dec_ab.py c-blosc_amd/libblosc_amd.so gpurun_tune_u16.so gpurun_tune_u32.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_dec_ab.txt
