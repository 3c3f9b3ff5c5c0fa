#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== dec zstd 8 GiB"
for d in bench19 linspace randwalk; do
  for m in 0 2; do
    echo -n "ZSTD2=$m "; BLOSC_AMD_ZSTD2=$m CODEC=zstd CLEVEL=3 CHUNKS=128 DATA=$d timeout 200 python scripts/dec_sweep.py 2>&1 | grep data=
  done
done | tee gpurun_out/n_dec_zstd.log
