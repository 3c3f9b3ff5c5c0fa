#!/bin/bash
# Round 2, call g: unshuffle with register-resident small-period patterns (A/B against the previous library), phase profiles
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_spans.py tests/test_gpu_decompress.py tests/test_gpu_baseline_geometry.py tests/test_gpu_modes.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x 2>&1 | tail -5 | tee gpurun_out/g_tests.log
echo "== dec A/B"
for d in bench19 linspace zeros randwalk; do
  for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so; do
    echo -n "$lib: "; BLOSC_AMD_LIB=$PWD/$lib DATA=$d timeout 100 python scripts/dec_sweep.py 2>&1 | grep data=
  done
done | tee gpurun_out/g_dec_ab.log
echo "== enc phase"; timeout 120 python scripts/enc_phase.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g_enc_phase.log
echo "== dec phase"; timeout 120 python scripts/dec_phase.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g_dec_phase.log
