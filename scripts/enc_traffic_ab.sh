cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for lib in c-blosc_amd/libblosc_amd.so gpurun_tune_w6.so gpurun_tune_s1.so gpurun_tune_par0.so; do
  for PMC in WRITE_SIZE FETCH_SIZE; do
    DATA=bench19 BLOSC_AMD_LIB=$PWD/$lib timeout 200 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmc_x -o pmc -- python scripts/enc_sweep.py > /tmp/pmc_x.log 2>&1
    f=$(find /tmp/pmc_x -name "*counter_collection.csv" | head -1)
    python - "$f" $PMC $lib <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if row['Counter_Name'] == sys.argv[2] and 'k_encode_streams' in row['Kernel_Name']: acc['enc'].append(float(row['Counter_Value']))
v=acc['enc']; print(sys.argv[3], sys.argv[2], 'launches', len(v), 'mean GB', sum(v)/len(v)*1024/1e9 if v else None)
PY
    rm -rf /tmp/pmc_x
  done
done
