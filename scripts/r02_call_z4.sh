#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2 3; do
  for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so; do
    echo -n "rep$rep lz4 T=8 shuffle $lib: "; BLOSC_AMD_LIB=$PWD/$lib DATA=bench19 timeout 100 python scripts/dec_sweep.py 2>&1 | grep data= | sed -e 's/.*k_decode_streams/k_decode_streams/'
  done
done | tee gpurun_out/z4_ab.log
for rep in 1 2; do
  for lib in gpurun_tune_base.so c-blosc_amd/libblosc_amd.so; do
    echo -n "rep$rep bench cfg2 $lib: "; BLOSC_AMD_LIB=$PWD/$lib timeout 200 python bench.py --config 2 --no-cpu-baseline --no-stock 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), {k:round(v['ms_avg'],3) for k,v in d['kernels'].items() if v['ms_avg']>0.05})"
  done
done | tee -a gpurun_out/z4_ab.log
