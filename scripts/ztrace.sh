#!/bin/bash
# kernel start / end times of one Zstd decompress call (reference-written config-4 frames), in launch order with their hardware queues
# (round 4: the timeline of the sliced multi-stream pipeline that was measured and not kept, profiles/r04/r04zp_*)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
LIB=${1:-c-blosc_amd/libblosc_amd.so}
CHUNKS=${CHUNKS:-128} DATA=bench19 CODEC=zstd CLEVEL=3 BLOSC_AMD_LIB=$PWD/$LIB timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ztrace -o t -- python scripts/dec_sweep.py > gpurun_out/ztrace.log 2>&1
f=$(find gpurun_out/ztrace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_zstd' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last call's kernels: everything after the last k_zstd_streams but one
ends = [i for i, r in enumerate(rows) if 'k_zstd_streams' in r['Kernel_Name']]
lo = ends[-2] + 1 if len(ends) > 1 else 0
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:ends[-1] + 1]:
    n = r['Kernel_Name'].split('(')[0].split('::')[-1][:22]
    print(f"{n:22s} queue {r.get('Queue_Id','?'):>3s}  start {(int(r['Start_Timestamp'])-t0)/1e6:8.3f} ms  end {(int(r['End_Timestamp'])-t0)/1e6:8.3f} ms")
PY
rm -rf gpurun_out/ztrace
