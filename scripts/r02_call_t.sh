#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_blpk.py tests/test_gpu_spans.py tests/test_gpu_getitem_batch.py -m gpu -q --no-header -p no:cacheprovider --timeout 300 -x 2>&1 | tail -8 | tee gpurun_out/t_tests.log
echo "== dec variants"
for d in bench19 linspace; do
  for lib in c-blosc_amd/libblosc_amd.so gpurun_tune_DST2.so c-blosc_amd/libblosc_amd.so; do
    echo -n "$lib: "; BLOSC_AMD_LIB=$PWD/$lib DATA=$d timeout 100 python scripts/dec_sweep.py 2>&1 | grep data=
  done
done | tee gpurun_out/t_dec_variants.log
