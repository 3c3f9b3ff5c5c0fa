#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_decompress.py tests/test_gpu_filters.py tests/test_gpu_getitem_batch.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x 2>&1 | tail -6 | tee gpurun_out/x_tests.log
timeout 600 python -m pytest tests/test_gpu_baseline_geometry.py tests/test_gpu_modes.py -m gpu -q --no-header -p no:cacheprovider --timeout 600 -x -k "bitshuffle or BLOCKDEC or defaults or FUSE" 2>&1 | tail -4 | tee -a gpurun_out/x_tests.log
echo "== cfg3 A/B"
for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so; do
  echo "$lib"; BLOSC_AMD_LIB=$PWD/$lib timeout 200 python bench.py --config 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), {k:round(v['ms_avg'],2) for k,v in d['kernels'].items() if v['ms_avg']>0.05}, {k:round(v,2) for k,v in d.get('decompress_stock_chunks',{}).get('kernels_ms',{}).items() if v>0.05})"
done | tee gpurun_out/x_cfg3_ab.log
