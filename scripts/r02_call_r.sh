#!/bin/bash
# SQ instruction counters over the encode kernel (bench19, LZ4): is it issue-bound?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export DATA=bench19 CHUNKS=128 CODECS=lz4
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_ANY SQ_INSTS_VALU_TRANS SQ_INSTS_FLAT SQ_INSTS_GDS"; do
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
  tail -2 gpurun_out/encpmc_$i.log
  rm -rf gpurun_out/encpmc_$i
done 2>&1 | tee gpurun_out/r_enc_pmc.log
