#!/bin/bash
# Round 2, call h: encoder with register-based backward extension (A/B against the previous library)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_compress.py tests/test_gpu_baseline_geometry.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x 2>&1 | tail -5 | tee gpurun_out/h_tests.log
echo "== enc A/B"
for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so; do
  echo "$lib"; BLOSC_AMD_LIB=$PWD/$lib CODECS=lz4,blosclz,zstd timeout 200 python scripts/enc_sweep.py 2>&1 | grep data=
done | tee gpurun_out/h_enc_ab.log
echo "== enc phase"; timeout 120 python scripts/enc_phase.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/h_enc_phase.log
