#!/bin/bash
# SQ counter passes over the decode kernels of scripts/dec_sweep.py (default: CODEC=zstd, reference-written frames, kernels k_zstd*: what do k_zstd_seq and
# k_zstd_exec wait for?; CODEC=lz4 CLEVEL=5 KFILTER=k_decode for the LZ4 kernel)
# (counters only with --kernel-trace: gpurun refuses other trace domains next to --pmc)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export DATA=${DATA:-bench19} CHUNKS=${CHUNKS:-128} CODEC=${CODEC:-zstd} CLEVEL=${CLEVEL:-3} KFILTER=${KFILTER:-k_zstd}
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/zpmc_$i -o pmc -- python scripts/dec_sweep.py > gpurun_out/zpmc_$i.log 2>&1
  f=$(find gpurun_out/zpmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get('Kernel_Name', '?').split('(')[0]
        if os.environ.get('KFILTER', 'k_zstd') not in k: continue
        acc[k.split('::')[-1][:24]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in sorted(acc.items()):
    for c, v in d.items(): print(f"{k:26s} {c:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
  rm -rf gpurun_out/zpmc_$i
done
