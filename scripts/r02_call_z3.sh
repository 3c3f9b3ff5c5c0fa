#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "blosclz 8 1 bench19" "blosclz 4 2 bench19" "lz4 4 2 bench19" "lz4 4 2 arange" "lz4 8 1 bench19" "lz4 8 1 randwalk" "lz4 8 1 linspace"; do
  set -- $cfg
  for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so gpurun_tune_MR6.so; do
    echo -n "$1 T=$2 shuffle=$3 $lib: "; CODEC=$1 BLOSC_AMD_LIB=$PWD/$lib TYPESIZE=$2 SHUFFLE=$3 DATA=$4 timeout 100 python scripts/dec_sweep.py 2>&1 | grep data= | sed -e 's/decompress kernels//' -e 's/k_decode_plan [0-9.]*//'
  done
done | tee gpurun_out/z3_dec_ab.log
