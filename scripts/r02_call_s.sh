#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== dec variants"
for d in bench19 linspace; do
  for lib in c-blosc_amd/libblosc_amd.so gpurun_tune_DST2.so gpurun_tune_LDNT.so gpurun_tune_BOTH.so c-blosc_amd/libblosc_amd.so; do
    echo -n "$lib: "; BLOSC_AMD_LIB=$PWD/$lib DATA=$d timeout 100 python scripts/dec_sweep.py 2>&1 | grep data=
  done
done | tee gpurun_out/s_dec_variants.log
