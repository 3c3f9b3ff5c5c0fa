#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05i
CHUNK_MIB=64 DATA=linspace,bench19,randwalk,smallints,arange,zeros,random timeout 900 python scripts/parity_hunt.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_parity_hunt_64MiB.txt | tail -12
CHUNK_MIB=16 CODECS=lz4hc,zlib,zstd CLEVELS=5,1,7 DATA=linspace,bench19,randwalk,smallints timeout 1200 python scripts/parity_hunt.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_parity_hunt_entropy.txt | tail -12
