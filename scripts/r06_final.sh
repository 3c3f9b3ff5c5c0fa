#!/bin/bash
# round 6, final pass on the round's last kernel sources: whole GPU suite, default bench line (its extra file is committed under the name the line prints),
# rocprofv3 kernel stats of config 2 / 3 with OWN chunks only (--no-stock) and, apart, of the decode of REFERENCE-WRITTEN chunks alone (scripts/dec_sweep.py), so
# that both averages can be read off rocprof by themselves, FETCH / WRITE passes of both.  Then, locally: scripts/make_traffic_json.py r06 2 ; ... r06 3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=${T:-r06_final}
if [ "${SKIP_SUITE:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${T}_pytest_gpu.log | tail -6
fi
echo "== default bench"; timeout 500 python bench.py --steps 20 --warmup 5 2> gpurun_out/${T}_bench_default.err > gpurun_out/${T}_bench_default.json; wc -c gpurun_out/${T}_bench_default.json gpurun_out/${T}_bench_default.err; cut -c1-700 gpurun_out/${T}_bench_default.json
cp gpurun_out/bench_extra.json gpurun_out/${T}_bench_extra.json 2>/dev/null
echo "== config 5 share"; timeout 500 python bench.py --chunks 512 --no-extra --no-cpu-baseline 2> gpurun_out/${T}_bench_512.err > gpurun_out/${T}_bench_512.json; cut -c1-400 gpurun_out/${T}_bench_512.json
for cfg in 2 3; do echo "== profile cfg $cfg"; CFG=$cfg TAG=r06 NOSTOCK=1 timeout 700 bash scripts/profile_config.sh 2>&1 | tail -25; done
echo "== stock decode alone under the kernel trace"
for spec in bench19:1:8:cfg2 bench19:2:4:cfg3; do
  IFS=: read d sh ts name <<< "$spec"
  DATA=$d SHUFFLE=$sh TYPESIZE=$ts timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_${name}_stock_trace -o trace -- python scripts/dec_sweep.py > gpurun_out/r06_${name}_stock_decode_under_trace.log 2>&1
  f=$(find gpurun_out/r06_${name}_stock_trace -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r06_${name}_stock_decode_kernel_stats.csv && head -5 "$f" | cut -c1-160
  rm -rf gpurun_out/r06_${name}_stock_trace
done
echo "== stock decode traffic"; TAG=r06 timeout 400 bash scripts/gpu_call.sh dectraffic 2>&1 | tail -8
