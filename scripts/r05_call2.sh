#!/bin/bash
# round 5, call 2: fused bit(un)shuffle of typesize 8 on the device, lazy input-ring variants, bench leg 3e
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05b
timeout 900 python -m pytest tests/test_gpu_bitunshuffle_fused.py tests/test_gpu_filters.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${T}_pytest.log | tail -6
echo "== decab"; ROUNDS=3 DECSETS="bench19:1:8 linspace:1:8 bench19:2:4 bench19:1:2" timeout 500 python scripts/dec_ab.py c-blosc_amd/libblosc_amd.so gpurun_tune_lazy1.so gpurun_tune_lazy2.so gpurun_tune_lazy4.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_dec_ab.txt
echo "== bench 3e / 3"; for c in 3e 3; do timeout 300 python bench.py --config $c --no-cpu-baseline --steps 5 --warmup 2 2> gpurun_out/${T}_bench_cfg$c.err | tee gpurun_out/${T}_bench_cfg$c.json | cut -c1-1500; done
