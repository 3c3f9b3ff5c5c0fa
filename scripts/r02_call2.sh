#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== quick"; timeout 180 python tests/gpu_quick.py 2>&1 | tee gpurun_out/quick.log | grep -v "True.*True" | tail -15
echo "== pytest subset"; timeout 900 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_spans.py tests/test_gpu_getitem_batch.py tests/test_gpu_compress.py -m gpu -q -x --no-header -p no:cacheprovider --timeout 300 2>&1 | tee gpurun_out/pytest_sub.log | tail -15
echo "== geometry"; timeout 600 python -m pytest tests/test_gpu_baseline_geometry.py -m gpu -q -x --no-header -p no:cacheprovider --timeout 300 -k "lz4-shuffle" 2>&1 | tee gpurun_out/pytest_geo.log | tail -8
echo "== bench cfg 2"; timeout 300 python bench.py --config 2 --no-cpu-baseline 2> gpurun_out/bench2.err | tee gpurun_out/bench2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:round(v['ms_avg'],3) for k,v in d['kernels'].items()}, d['decompress_stock_chunks']['kernels_ms'])"
tail -3 gpurun_out/bench2.err
