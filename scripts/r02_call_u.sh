#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== enc variants"
for v in "base:c-blosc_amd/libblosc_amd.so:32" "shnt:gpurun_tune_SHNT.so:32" "la8:c-blosc_amd/libblosc_amd.so:8" "la4:c-blosc_amd/libblosc_amd.so:4" "la96:c-blosc_amd/libblosc_amd.so:96" "base:c-blosc_amd/libblosc_amd.so:32"; do
  IFS=: read n lib la <<< "$v"
  echo "$n"; BLOSC_AMD_ENC_LOOKAHEAD=$la BLOSC_AMD_LIB=$PWD/$lib CODECS=lz4 DATA=bench19,linspace,randwalk timeout 200 python scripts/enc_sweep.py 2>&1 | grep data=
done | tee gpurun_out/u_enc_variants.log
