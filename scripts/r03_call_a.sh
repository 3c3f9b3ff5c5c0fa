#!/bin/bash
# Round 3, first call: what the end of round 2 built without GPU time left to run it.
#   1. the GPU tests of the new encoder options (Zstd per-block tables, the LZ4HC-grade search in front of the Zstd / zlib writers) - they
#      have only run on the wavefront emulator so far (tests/test_wave_emu_encoders.py);
#   2. bench lines for them next to their baselines: 4 / 4t / 4s (Zstd: predefined tables / per-block tables / tables + search),
#      z / zs / zd (zlib: plain / search / dynamic codes + search), h ("lz4hc" clevel 9), 4r / 4h (Zstd on noisy small integers: raw / Huffman-coded literals), and the
#      headline config 2 as the session's reference point.
# Everything lands in gpurun_out/ (copy what is kept to profiles/r03a_*).  About 4 GPU-minutes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle oracle > /dev/null 2>&1
echo "== new GPU tests"; timeout 300 python -m pytest tests/test_gpu_zstd_tables.py tests/test_gpu_lz4hc.py -m gpu -q --no-header -p no:cacheprovider -s 2>&1 | tee gpurun_out/a_pytest_new.log | grep -v "^$" | tail -25
for cfg in 2 4 4t 4s z zs zd h 4r 4h; do
  echo "== bench cfg $cfg"
  timeout 200 python bench.py --config $cfg --no-cpu-baseline --steps 5 --warmup 2 2> gpurun_out/a_bench_cfg$cfg.err | tee gpurun_out/a_bench_cfg$cfg.json | cut -c1-220
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/a_bench_cfg*.json")):
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line); c = d["config"]
            print(f"{c['name']:3s} {d['value']:8.1f} {d['unit']}  ms/step {d['ms_per_step']:7.2f}  " + "  ".join(f"{k}={c[k]}" for k in c if k.startswith(("ratio", "compress_ms", "decompress_ms"))))
PY
