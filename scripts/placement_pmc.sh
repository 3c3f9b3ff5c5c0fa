#!/bin/bash
# The decode kernel's two placement states under the L2 / EA / TCP / SQ counters: ONE process per pass walks a list of arena offsets
# (scripts/placement_skew.py), rocprofv3 collects the counters and the duration of every k_decode_streams launch, and the launches are split at the
# middle of the duration range: which counter differs between the slow and the fast launches?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export SKEWS="${SKEWS:-0 4 8 16 20 36 48 96 384 1024}"
out=gpurun_out/${TAG:-r05y}_placement_states_pmc.txt; [ -n "${APPEND:-}" ] || : > $out
i=0
for PMC in \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum" \
  "TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_IB_STALL_sum" \
  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" \
  "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
  "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
  "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
  "GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_TC_BUSY GRBM_GUI_ACTIVE" ; do      # group 9 ran into its timeout on the pool (round 5): only with PASSES=9
  i=$((i+1)); [ -n "${PASSES:-}" ] && ! echo " $PASSES " | grep -q " $i " && continue
  [ -z "${PASSES:-}" ] && [ $i = 9 ] && continue
  rm -rf gpurun_out/ppmc_$i
  timeout 170 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/ppmc_$i -o pmc -- python scripts/placement_skew.py > gpurun_out/ppmc_$i.log 2>&1
  echo "== pass $i: $PMC  (rc $?)" >> $out
  python scripts/placement_pmc_split.py gpurun_out/ppmc_$i >> $out 2>&1
  rm -rf gpurun_out/ppmc_$i
done
cat $out
