#!/bin/bash
# Round 3, call b: (1) scratch-ring micro-benchmark (does a bounded ring keep the plane-major round trip out of HBM?),
# (2) k_decode_streams on reference-written bench19 chunks: waves per CU x queue order, (3) its phase profile,
# (4) r03_call_a.sh (device timing of the round-2 encoder options).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== ring micro"; timeout 120 scripts/micro/ring 2>&1 | tee gpurun_out/b_ring_micro.txt
echo "== decode sweep (stock bench19 chunks)"
for wpc in 24 20 16 12; do for sched in 1 0; do
  echo -n "WPC=$wpc SCHED=$sched  "; BLOSC_AMD_DEC_WPC=$wpc BLOSC_AMD_SCHED=$sched timeout 120 python scripts/dec_sweep.py 2>&1 | tail -1
done; done | tee gpurun_out/b_dec_sweep.txt
echo "== decode phases"; timeout 120 python scripts/dec_phase.py 2>&1 | tee gpurun_out/b_dec_phase.txt | tail -30
bash scripts/r03_call_a.sh
