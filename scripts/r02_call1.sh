#!/bin/bash
# Round 2, GPU call 1: sanity of the round-1 kernels after the host fixes, the new tests, the unmeasured configs,
# and counter passes over the two stream kernels.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== micro"; timeout 60 scripts/micro/lds_unaligned 2>&1 | tee gpurun_out/micro.log
echo "== quick"; timeout 180 python tests/gpu_quick.py 2>&1 | tee gpurun_out/quick.log | tail -8
echo "== pytest gpu"; timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --timeout 600 2>&1 | tee gpurun_out/pytest_gpu.log | tail -40
for cfg in 2 3 4; do
  echo "== bench cfg $cfg"; timeout 300 python bench.py --config $cfg 2> gpurun_out/bench_cfg$cfg.err | tee gpurun_out/bench_cfg$cfg.json | cut -c1-600
done
for cfg in 2b 2c 2d 3b 3c 1g 4b 4c; do
  echo "== bench cfg $cfg"; timeout 200 python bench.py --config $cfg --no-cpu-baseline 2> gpurun_out/bench_cfg$cfg.err | tee gpurun_out/bench_cfg$cfg.json | cut -c1-400
done
echo "== decode PMC"
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/decpmc_$i -o pmc -- python scripts/dec_sweep.py > gpurun_out/decpmc_$i.log 2>&1
  f=$(find gpurun_out/decpmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a gpurun_out/decpmc_summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get('Kernel_Name','?').split('(')[0][-40:]
        if 'decode_streams' not in k: continue
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    for c, v in d.items(): print(f"{c:44s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
  rm -rf gpurun_out/decpmc_$i
done
