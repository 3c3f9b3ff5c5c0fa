#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel trace of `bench.py --config $CFG`, then FETCH_SIZE / WRITE_SIZE passes (one
# counter per pass, --kernel-trace only) of a one-step run of the same workload.   usage: CFG=2 TAG=r02 bash scripts/profile_config.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CFG=${CFG:-2}; TAG=${TAG:-r05}_cfg${CFG}
# which kernel sources these passes ran on (bench.py quotes a traffic figure only for the sources it runs itself)
python -c "import bench; print(bench.src_fingerprint())" > gpurun_out/${TAG}_src_fingerprint.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o trace -- python bench.py --config $CFG --no-cpu-baseline --no-extra ${NOSTOCK:+--no-stock} > gpurun_out/${TAG}_bench_under_trace.log 2>&1      # NOSTOCK=1: own chunks only, the decode of reference-written chunks is traced apart (scripts/r06_final.sh)
grep '^{' gpurun_out/${TAG}_bench_under_trace.log | tail -1 > gpurun_out/${TAG}_bench_under_trace.json
f=$(find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats.csv && head -14 "$f" | cut -c1-160
rm -rf gpurun_out/${TAG}_trace
for PMC in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_$PMC -o pmc -- python bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-stock --no-extra > gpurun_out/${TAG}_pmc_$PMC.log 2>&1
  f=$(find gpurun_out/${TAG}_pmc_$PMC -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $PMC gpurun_out/${TAG}_pmc_$PMC.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list); grid = {}
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if row['Counter_Name'] != sys.argv[2]: continue
        k = row['Kernel_Name'].split('(')[0]
        k = k[-60:]
        acc[k].append(float(row['Counter_Value'])); grid[k] = row['Grid_Size']
with open(sys.argv[3], 'w') as out:
    out.write("kernel,counter,launches,mean_value_KB,max_value_KB,grid_size\n")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        out.write(f"\"{k}\",{sys.argv[2]},{len(v)},{sum(v)/len(v):.3f},{max(v):.3f},{grid[k]}\n")
print(open(sys.argv[3]).read()[:1500])
PY
  rm -rf gpurun_out/${TAG}_pmc_$PMC
done
