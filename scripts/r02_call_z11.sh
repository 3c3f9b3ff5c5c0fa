#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_spans.py tests/test_gpu_modes.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x 2>&1 | tail -2
for lib in gpurun_tune_BLZOFF.so c-blosc_amd/libblosc_amd.so; do
  echo -n "blosclz T=8 shuffle $lib: "; CODEC=blosclz BLOSC_AMD_LIB=$PWD/$lib DATA=bench19 timeout 100 python scripts/dec_sweep.py 2>&1 | grep data= | sed -e 's/.*k_decode_streams/k_decode_streams/'
done | tee gpurun_out/z11.log
for cfg in 1g 3; do timeout 200 python bench.py --config $cfg --no-cpu-baseline 2> /dev/null | tee gpurun_out/g_bench_cfg$cfg.json | cut -c1-130; done
