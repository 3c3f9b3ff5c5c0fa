#!/bin/bash
# One parametrised script for every `gpurun` call of a round (replaces the one-shot scripts/r02_call_*.sh of round 2):
#   gpurun --timeout 900 -- 'bash scripts/gpu_call.sh <stage> [<stage> ...]'
# Every stage runs under its own timeout and writes to gpurun_out/<tag>_*; copy what is kept to profiles/.
# TAG=<prefix> names the outputs (default: r03).  Same-session A/B: put the library to compare in gpurun_tune_base.so.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-r04}
make -C oracle oracle > /dev/null 2>&1

bench_cfg() {   # bench_cfg <cfg> [extra bench.py args]: one short JSON line without the CPU leg
  local cfg=$1; shift
  echo "== bench cfg $cfg"
  timeout 200 python bench.py --config "$cfg" --no-cpu-baseline --steps 5 --warmup 2 "$@" 2> gpurun_out/${TAG}_bench_cfg$cfg.err | tee gpurun_out/${TAG}_bench_cfg$cfg.json | cut -c1-240
}
summarise() {
  python - "$TAG" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(f"gpurun_out/{sys.argv[1]}_bench_cfg*.json")):
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line); c = d["config"]
            print(f"{c['name']:3s} {d['value']:8.1f} {d['unit']}  ms/step {d['ms_per_step']:7.2f}  " + "  ".join(f"{k}={c[k]}" for k in c if k.startswith(("ratio", "compress_ms", "decompress_ms"))))
PY
}

for stage in "$@"; do
  case $stage in
    quick)      timeout 300 python tests/gpu_quick.py 2>&1 | tail -15 ;;
    suite)      timeout 2400 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${TAG}_pytest_gpu.log | tail -15 ;;
    smoke)      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 ;;
    ring)       echo "== ring micro"; timeout 120 scripts/micro/ring 2>&1 | tee gpurun_out/${TAG}_ring_micro.txt ;;
    decphase)   timeout 120 python scripts/dec_phase.py 2>&1 | tee gpurun_out/${TAG}_dec_phase.txt | tail -30 ;;
    dec)        # decode-only timing of reference-written chunks: the library under test and every gpurun_tune_*.so next to it (same-session A/B)
                for spec in ${DECSETS:-bench19:1:8 linspace:1:8 randwalk:1:8 bench19:2:4}; do
                  IFS=: read d sh ts <<< "$spec"
                  for lib in c-blosc_amd/libblosc_amd.so gpurun_tune_*.so; do
                    [ -f $lib ] && { echo -n "$lib shuffle=$sh T=$ts "; DATA=$d SHUFFLE=$sh TYPESIZE=$ts BLOSC_AMD_LIB=$PWD/$lib timeout 150 python scripts/dec_sweep.py 2>&1 | tail -1; }
                  done
                done | tee -a gpurun_out/${TAG}_dec_ab.txt ;;
    decab)      # the same in ONE process, the builds taking turns on the same buffers (scripts/dec_ab.py): process-to-process differences cancel
                DECSETS="${DECSETS:-bench19:1:8 linspace:1:8 randwalk:1:8 bench19:2:4}" timeout 600 python scripts/dec_ab.py c-blosc_amd/libblosc_amd.so gpurun_tune_*.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_dec_ab1.txt ;;
    dectraffic) # FETCH_SIZE / WRITE_SIZE of the decode kernels (one counter per pass, --kernel-trace only), reference-written bench19 chunks
                for PMC in FETCH_SIZE WRITE_SIZE; do
                  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_$PMC -o pmc -- python scripts/dec_sweep.py > gpurun_out/${TAG}_pmc_$PMC.log 2>&1
                  f=$(find gpurun_out/${TAG}_pmc_$PMC -name "*counter_collection.csv" | head -1)
                  [ -n "$f" ] && python - "$f" $PMC <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if row['Counter_Name'] == sys.argv[2]: acc[row['Kernel_Name'].split('(')[0][-40:]].append(float(row['Counter_Value']))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:6]: print(f"{sys.argv[2]} {k}: launches {len(v)} mean {sum(v)/len(v)/1e6:.1f} M units  max {max(v)/1e6:.1f}")
PY
                  rm -rf gpurun_out/${TAG}_pmc_$PMC
                done | tee gpurun_out/${TAG}_dec_traffic.txt ;;
    hostthreads) for c in 1 4; do BLOSC_AMD_CONTEXTS=$c timeout 200 python scripts/host_abi_threads.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/${TAG}_host_abi_threads.txt ;;
    threads)    timeout 600 python -m pytest tests/test_gpu_threads.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${TAG}_pytest_threads.log | tail -5 ;;
    zstdtests)  timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_zstd_tables.py tests/test_gpu_zlib.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${TAG}_pytest_zstd.log | tail -5 ;;
    dectests)   timeout 900 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_spans.py tests/test_gpu_baseline_geometry.py tests/test_gpu_getitem_batch.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${TAG}_pytest_dec.log | tail -8 ;;
    enc)        for d in ${DATA:-bench19 linspace randwalk}; do for lib in c-blosc_amd/libblosc_amd.so gpurun_tune_*.so; do
                  [ -f $lib ] && { echo -n "$lib "; DATA=$d BLOSC_AMD_LIB=$PWD/$lib timeout 150 python scripts/enc_sweep.py 2>&1 | tail -1; }
                done; done | tee -a gpurun_out/${TAG}_enc_ab.txt ;;
    encab)      ENCSETS="${ENCSETS:-bench19:1:8 linspace:1:8 randwalk:1:8 bench19:2:4}" timeout 600 python scripts/enc_ab.py c-blosc_amd/libblosc_amd.so gpurun_tune_*.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_enc_ab1.txt ;;
    encphase)   timeout 200 python scripts/enc_phase.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_enc_phase.txt | tail -14 ;;
    encopts)    # device timing of the encoder options built at the end of round 2 (formerly scripts/r03_call_a.sh)
                timeout 300 python -m pytest tests/test_gpu_zstd_tables.py tests/test_gpu_lz4hc.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${TAG}_pytest_encopts.log | tail -5
                for cfg in 2 4 4t 4s z zs zd h 4r 4h; do bench_cfg $cfg; done; summarise ;;
    bench)      echo "== default bench line"; timeout 600 python bench.py 2> gpurun_out/${TAG}_bench_default.err | tee gpurun_out/${TAG}_bench_default.json | cut -c1-400 ;;
    benchall)   for cfg in ${CFGS:-2 2b 2c 2d 3 3b 3c 4 4b 4c 1g z zb}; do bench_cfg $cfg; done; summarise ;;
    profile)    for cfg in ${CFGS:-2}; do CFG=$cfg TAG=$TAG bash scripts/profile_config.sh; done ;;
    *)          echo "unknown stage $stage" ;;
  esac
done
