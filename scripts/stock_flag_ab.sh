for flags in "--no-stock" "" "--no-stock" ""; do
echo -n "flags[$flags]: "; python bench.py --no-extra --no-cpu-baseline --steps 6 $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v['ms_avg'], 3) for k, v in d['kernels'].items() if v['ms_avg'] > 1})"
done
