#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== pytest subset"; timeout 900 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_spans.py tests/test_gpu_getitem_batch.py tests/test_gpu_compress.py -m gpu -q --no-header -p no:cacheprovider --timeout 300 2>&1 | tee gpurun_out/pytest_sub.log | tail -15
echo "== bench cfg 2"; timeout 300 python -X faulthandler bench.py --config 2 --no-cpu-baseline > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "bench rc=$?"
tail -25 gpurun_out/bench2.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/bench2.json').read().strip().splitlines()[-1])
    print({k:round(v['ms_avg'],3) for k,v in d['kernels'].items()}, d['decompress_stock_chunks']['kernels_ms'])
except Exception as e: print('no json', e)
PY
