#!/bin/bash
# Round 2, call m: Zstd two-phase decode with the tables in a global scratch (BLOSC_AMD_ZSTD2=2) vs LDS (=1) vs one wave per frame (=0)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests mode 2"; BLOSC_AMD_ZSTD2=2 timeout 600 python -m pytest tests/test_gpu_zstd.py -m gpu -q --no-header -p no:cacheprovider --timeout 300 -x 2>&1 | tail -4 | tee gpurun_out/m_tests.log
echo "== dec zstd"
for d in bench19 linspace randwalk; do
  for m in 0 1 2; do
    echo -n "ZSTD2=$m "; BLOSC_AMD_ZSTD2=$m CODEC=zstd CLEVEL=3 CHUNKS=32 DATA=$d timeout 200 python scripts/dec_sweep.py 2>&1 | grep data=
  done
done | tee gpurun_out/m_dec_zstd.log
