#!/bin/bash
# round 5, call 4: the whole GPU suite on the round's code, the default bench line, the Zstd Huffman-literal option on the device
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
T=r05d
timeout 2400 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/${T}_pytest_gpu.log | tail -8
echo "== default bench"; timeout 600 python bench.py --steps 20 --warmup 5 2> gpurun_out/${T}_bench_default.err > gpurun_out/${T}_bench_default.json; wc -c gpurun_out/${T}_bench_default.json gpurun_out/${T}_bench_default.err; cat gpurun_out/${T}_bench_default.json
cp gpurun_out/bench_extra.json gpurun_out/${T}_bench_extra.json 2>/dev/null
echo "== zstd huffman"; for c in 4r 4h; do timeout 300 python bench.py --config $c --no-cpu-baseline --steps 3 --warmup 1 2> gpurun_out/${T}_bench_cfg$c.err | tee gpurun_out/${T}_bench_cfg$c.json | cut -c1-900; done
BLOSC_AMD_ZSTD_HUFFMAN=1 timeout 300 python bench.py --config 4 --no-cpu-baseline --steps 3 --warmup 1 2> gpurun_out/${T}_bench_cfg4_huf.err | tee gpurun_out/${T}_bench_cfg4_huf.json | cut -c1-900
