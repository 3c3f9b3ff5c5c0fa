#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_spans.py tests/test_gpu_compress.py tests/test_gpu_modes.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x 2>&1 | tail -4 | tee gpurun_out/z2_tests.log
timeout 600 python -m pytest tests/test_gpu_baseline_geometry.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x -k "lz4 or blosclz" 2>&1 | tail -3 | tee -a gpurun_out/z2_tests.log
echo "== dec A/B (stock BloscLZ chunks)"
for cfg in "4 2 bench19" "8 1 bench19"; do
  set -- $cfg
  for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so; do
    echo -n "T=$1 shuffle=$2 $lib: "; CODEC=blosclz BLOSC_AMD_LIB=$PWD/$lib TYPESIZE=$1 SHUFFLE=$2 DATA=$3 timeout 100 python scripts/dec_sweep.py 2>&1 | grep data=
  done
done | tee gpurun_out/z2_dec_ab.log
