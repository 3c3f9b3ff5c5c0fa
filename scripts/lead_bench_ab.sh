for r in 1 2; do for lib in c-blosc_amd/libblosc_amd.so gpurun_tune_lead64.so gpurun_tune_lead256.so; do for c in 2 3; do
echo -n "$lib cfg$c: "; BLOSC_AMD_LIB=$PWD/$lib python bench.py --config $c --no-extra --no-cpu-baseline --steps 8 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['kernels']['k_encode_streams']['ms_avg'],2), 'own dec', round(d['kernels']['k_decode_streams']['ms_avg'],3), 'stock', round(d['decompress_stock_chunks']['kernels_ms']['k_decode_streams'],3))"
done; done; done
