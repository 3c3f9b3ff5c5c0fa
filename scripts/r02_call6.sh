#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
timeout 180 python tests/gpu_quick.py 2>&1 | grep -E "filters|shuffle=2" | head -20
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_filters.py tests/test_gpu_compress.py tests/test_gpu_decompress.py -m gpu -q -x --no-header -p no:cacheprovider --timeout 300 2>&1 | tail -5
for cfg in 3 3c; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline 2> gpurun_out/bench_cfg$cfg.err | grep '^{' > gpurun_out/bench_cfg$cfg.json
  python - $cfg <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/bench_cfg{sys.argv[1]}.json').read())
print(sys.argv[1], 'value',round(d['value'],1),'ratio',round(d['ratio'],2),{k:round(v['ms_avg'],3) for k,v in d['kernels'].items()}, d['decompress_stock_chunks']['kernels_ms'] if d['decompress_stock_chunks'] else None)
PY
done
