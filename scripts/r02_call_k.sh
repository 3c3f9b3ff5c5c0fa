#!/bin/bash
# Round 2, call k: Zlib encode on the GPU (tests), A/B of the per-codec encode kernels against the previous library
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_zlib.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x 2>&1 | tail -12 | tee gpurun_out/k_tests.log
echo "== enc A/B"
for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so; do
  echo "$lib"; BLOSC_AMD_LIB=$PWD/$lib CODECS=lz4,blosclz,zstd DATA=bench19,randwalk timeout 200 python scripts/enc_sweep.py 2>&1 | grep data=
done | tee gpurun_out/k_enc_ab.log
echo "== zlib enc"; CODECS=zlib timeout 200 python scripts/enc_sweep.py 2>&1 | grep data= | tee gpurun_out/k_enc_zlib.log
