#!/bin/bash
# rocprofv3 kernel trace + PMC passes of a reduced bench; summaries land in gpurun_out/prof_*
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CH=${CHUNKS:-32}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -o trace -- python bench.py --chunks $CH --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_trace.log 2>&1
tail -3 gpurun_out/prof_trace.log | cut -c1-600
find gpurun_out/prof_trace -name "*kernel_stats*" | head -3
for f in $(find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -1); do cat $f | cut -c1-220; done
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $PMC | cut -d' ' -f1)
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/prof_pmc_$tag -o pmc -- python bench.py --chunks $CH --steps 1 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/prof_pmc_$tag.log 2>&1
  f=$(find gpurun_out/prof_pmc_$tag -name "*counter_collection.csv" | head -1)
  echo "== $PMC -> $f"
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get('Kernel_Name','?').split('(')[0][-40:]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    print(k, {c: (sum(v)/len(v), len(v)) for c, v in d.items()})
PY
done
