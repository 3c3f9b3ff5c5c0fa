#!/bin/bash
# Fused (un)shuffle of the "other" typesizes against the three-pass form (BLOSC_AMD_FUSE=0), same box, same data.
# usage (on the GPU box): bash scripts/generic_typesize_ab.sh "3 6 12 24" "linspace random" > gpurun_out/xx.txt
for D in ${2:-linspace random}; do for T in ${1:-3 6 12 24}; do for F in 1 0; do
  echo "== data=$D T=$T FUSE=$F"
  BLOSC_AMD_FUSE=$F python bench.py --config 2 --typesize $T --data $D --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c '
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
s = d.get("decompress_stock_chunks") or {}
print(d["value"], d["kernels"], "stock", s.get("kernels_ms"), "ratio", d.get("ratio"))'
done; done; done
