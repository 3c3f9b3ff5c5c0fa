"""Host logic without a GPU: the queue builders of the persistent kernels (c-blosc_amd/csrc/queue_order.h)
are plain C++; tests/tools/sched_check.cpp compiles them with g++ and checks the invariants the kernels rely
on (every stream exactly once on its block's XCD, shuffle task before the block's streams, offsets a prefix
sum, expensive planes kept out of the queue tails) over 400 random batch geometries; and, since round 3, that mixed batches are
partitioned: Zstd blocks in nobody's queue, zlib blocks in the zlib kernel's queues and only there, the rest in k_decode_streams'."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_queue_builders_invariants():
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "sched_check")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "c-blosc_amd", "csrc"),
                               os.path.join(ROOT, "tests", "tools", "sched_check.cpp"), "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "sched_check OK" in out.stdout, out.stdout + out.stderr
