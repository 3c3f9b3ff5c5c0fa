#!/bin/bash
# Round 2, call i: raw planes read in place by the fused unshuffle (A/B against the previous library)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_spans.py tests/test_gpu_decompress.py tests/test_gpu_baseline_geometry.py tests/test_gpu_modes.py tests/test_gpu_getitem_batch.py tests/test_gpu_compress.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x 2>&1 | tail -5 | tee gpurun_out/i_tests.log
echo "== dec A/B"
for d in randwalk bench19 linspace; do
  for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so; do
    echo -n "$lib: "; BLOSC_AMD_LIB=$PWD/$lib DATA=$d timeout 100 python scripts/dec_sweep.py 2>&1 | grep data=
  done
done | tee gpurun_out/i_dec_ab.log
