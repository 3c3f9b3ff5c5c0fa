#!/bin/bash
# round 5, call 3: RCCL exchange from C, FETCH_SIZE calibration for the encoder's access shapes, SQ counters of the two hot kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05c
timeout 600 python -m pytest tests/test_gpu_rccl_exchange.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${T}_pytest.log | tail -12
echo "== fetch calibration"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_calib -o pmc -- scripts/micro/fetch_calib > gpurun_out/${T}_fetch_calib.txt 2>&1
f=$(find gpurun_out/${T}_calib -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" >> gpurun_out/${T}_fetch_calib.txt <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if row['Counter_Name'] == 'FETCH_SIZE' and row['Kernel_Name'].startswith('k_'): print(f"FETCH_SIZE {row['Kernel_Name'].split('(')[0]:12s} {float(row['Counter_Value']):14.0f} KiB (as reported)")
PY
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/${T}_calib2 -o pmc -- scripts/micro/fetch_calib > /dev/null 2>&1
f=$(find gpurun_out/${T}_calib2 -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" >> gpurun_out/${T}_fetch_calib.txt <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if row['Kernel_Name'].startswith('k_'): print(f"{row['Counter_Name']:24s} {row['Kernel_Name'].split('(')[0]:12s} {float(row['Counter_Value']):14.0f}")
PY
rm -rf gpurun_out/${T}_calib gpurun_out/${T}_calib2
grep -v amdgpu.ids gpurun_out/${T}_fetch_calib.txt | tail -30
echo "== SQ counters: decode of reference-written chunks (config 2)"
CODEC=lz4 CLEVEL=5 KFILTER=k_decode_streams bash scripts/zstd_pmc.sh 2>&1 | tee gpurun_out/${T}_lz4_decode_sq_counters.txt | tail -30
echo "== SQ counters: encode (config 2)"
bash scripts/enc_pmc.sh 2>&1 | tee gpurun_out/${T}_encode_sq_counters.txt | tail -40
