#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for so in c-blosc_amd/libblosc_amd.so gpurun_tune_*.so; do
  for wpc in 16 24 32; do
    BLOSC_AMD_LIB=$PWD/$so BLOSC_AMD_ENC_WPC=$wpc BLOSC_AMD_DEC_WPC=$wpc timeout 90 python bench.py --chunks 64 --steps 2 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print('$so wpc=$wpc', 'enc %.2f dec %.2f stockdec %.2f' % (k['k_encode_streams']['ms_avg'], k['k_decode_streams']['ms_avg'], d['decompress_stock_chunks']['k_decode_streams_ms']))"
  done
done
