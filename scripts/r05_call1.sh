#!/bin/bash
# round 5, call 1: the harness changes on the device (bench line, new tests), leave-one-out / deeper-unshuffle A/B, typesize-2 phase profiles
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05a
timeout 600 python -m pytest tests/test_gpu_multigpu_nccl.py tests/test_gpu_multi_entry.py tests/test_gpu_threads.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${T}_pytest.log | tail -6
echo "== default bench"; timeout 600 python bench.py --steps 20 --warmup 5 2> gpurun_out/${T}_bench_default.err > gpurun_out/${T}_bench_default.json; wc -c gpurun_out/${T}_bench_default.json gpurun_out/${T}_bench_default.err; cut -c1-600 gpurun_out/${T}_bench_default.json
cp gpurun_out/bench_extra.json gpurun_out/${T}_bench_extra.json 2>/dev/null
echo "== decab"; NOCHECK=1 ROUNDS=3 DECSETS="bench19:1:8 linspace:1:8 bench19:1:2 bench19:1:4" timeout 500 python scripts/dec_ab.py c-blosc_amd/libblosc_amd.so gpurun_tune_u16.so gpurun_tune_u32.so gpurun_tune_loo26.so gpurun_tune_loo15.so gpurun_tune_loo1256.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_dec_ab.txt
echo "== phases T=2"; TYPESIZE=2 timeout 150 python scripts/dec_phase.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_dec_phase_t2.txt | head -12
TYPESIZE=2 timeout 150 python scripts/enc_phase.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_enc_phase_t2.txt | head -8
TYPESIZE=8 timeout 150 python scripts/enc_phase.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_enc_phase_t8.txt | head -12
