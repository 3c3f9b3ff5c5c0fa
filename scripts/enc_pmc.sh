#!/bin/bash
# PMC passes over the encode kernel only (scripts/enc_sweep.py, one dataset)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export DATA=${DATA:-bench19} CHUNKS=${CHUNKS:-128}
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/encpmc_$i -o pmc -- python scripts/enc_sweep.py > gpurun_out/encpmc_$i.log 2>&1
  f=$(find gpurun_out/encpmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get('Kernel_Name','?').split('(')[0][-40:]
        if 'encode_streams' not in k: continue
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    for c, v in d.items(): print(f"{c:44s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
  rm -rf gpurun_out/encpmc_$i
done
