#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 600 2>&1 | tee gpurun_out/pytest_gpu.log | tail -15
for cfg in 4 4b; do
  echo "== bench cfg $cfg"; timeout 300 python bench.py --config $cfg --no-cpu-baseline 2> gpurun_out/bench_cfg$cfg.err | grep '^{' > gpurun_out/bench_cfg$cfg.json
  python - $cfg <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/bench_cfg{sys.argv[1]}.json').read())
print(sys.argv[1], 'value',round(d['value'],1),'ratio',round(d['ratio'],2),{k:round(v['ms_avg'],3) for k,v in d['kernels'].items()}, 'stock ratio', d['decompress_stock_chunks']['ratio'] if d['decompress_stock_chunks'] else None)
PY
done
