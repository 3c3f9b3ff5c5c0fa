#!/bin/bash
# instruction-cache counters of the two hot kernels (config 2): are 16 - 24 waves per CU at different places of a 40 - 60 KB kernel bound by instruction fetch?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for what in dec enc; do
  script=scripts/dec_sweep.py; filt=k_decode_streams; [ $what = enc ] && { script=scripts/enc_sweep.py; filt=k_encode_streams; export DATA=bench19; }
  i=0
  for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
    i=$((i+1))
    CODEC=lz4 CLEVEL=5 timeout 150 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/icpmc_$i -o pmc -- python $script > gpurun_out/icpmc_$i.log 2>&1
    f=$(find gpurun_out/icpmc_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && KF=$filt python - "$f" <<'PY'
import csv, sys, collections, os
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if os.environ['KF'] in row.get('Kernel_Name', ''): acc[row['Counter_Name']].append(float(row['Counter_Value']))
for c, v in acc.items(): print(f"{os.environ['KF']:18s} {c:30s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
    rm -rf gpurun_out/icpmc_$i
  done
done
