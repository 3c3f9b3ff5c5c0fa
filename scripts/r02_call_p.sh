#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_compress.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x -k "zstd" 2>&1 | tail -6 | tee gpurun_out/p_tests.log
echo "== enc A/B"
for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so; do
  echo "$lib"; BLOSC_AMD_LIB=$PWD/$lib CODECS=zstd CLEVEL=3 timeout 200 python scripts/enc_sweep.py 2>&1 | grep data=
done | tee gpurun_out/p_enc_ab.log
