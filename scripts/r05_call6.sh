#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05f
for spec in linspace:1:4 linspace:1:2 randwalk:1:16 linspace:1:16; do IFS=: read d sh ts <<< "$spec"; DATA=$d SHUFFLE=$sh TYPESIZE=$ts timeout 300 python scripts/dbg_case.py c-blosc_amd/libblosc_amd.so gpurun_tune_g1.so gpurun_tune_r04.so 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/${T}_dbg_case.txt
