#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_zlib.py tests/test_gpu_modes.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x -k "zstd or zlib" 2>&1 | tail -3 | tee gpurun_out/z6_tests.log
timeout 600 python -m pytest tests/test_gpu_baseline_geometry.py tests/test_gpu_compress.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x -k "zstd or zlib" 2>&1 | tail -3 | tee -a gpurun_out/z6_tests.log
echo "== dec A/B"
for c in zstd zlib; do for d in bench19 randwalk; do
  for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so; do
    echo -n "$c $d $lib: "; CODEC=$c CLEVEL=$([ $c = zstd ] && echo 3 || echo 5) BLOSC_AMD_LIB=$PWD/$lib DATA=$d timeout 100 python scripts/dec_sweep.py 2>&1 | grep data= | sed -e 's/.*ms\/call//'
  done
done; done | tee gpurun_out/z6_dec_ab.log
