"""GPU: the fallback switches of the engine are part of the product, so they are part of the
suite: BLOSC_AMD_SINGLE_QUEUE (one task queue, stand-alone filters: what a partitioned device gets),
BLOSC_AMD_FUSE (filters in kernels of their own), BLOSC_AMD_SPANS (periodic planes through the scratch),
BLOSC_AMD_SCHED (plain block order).  The switches are read
once per process, so every combination runs tests/tools/mode_check.py in a process of its own."""
import itertools
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWITCHES = ["BLOSC_AMD_SINGLE_QUEUE", "BLOSC_AMD_FUSE", "BLOSC_AMD_SPANS", "BLOSC_AMD_SCHED"]
DEFAULTS = {"BLOSC_AMD_SINGLE_QUEUE": "0", "BLOSC_AMD_FUSE": "1", "BLOSC_AMD_SPANS": "1", "BLOSC_AMD_SCHED": "1"}


def _combos():
    # every single switch flipped, every pair flipped, and everything flipped at once
    seen = []
    for k in range(0, 3):
        for flip in itertools.combinations(SWITCHES, k):
            seen.append(flip)
    seen.append(tuple(SWITCHES))
    return seen


@pytest.mark.parametrize("flip", _combos(), ids=lambda f: "+".join(s.replace("BLOSC_AMD_", "") for s in f) or "defaults")
def test_mode_combination(flip):
    env = dict(os.environ)
    for s in SWITCHES:
        v = DEFAULTS[s]
        if s in flip:
            v = "1" if v == "0" else "0"
        env[s] = v
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "mode_check.py")], env=env, timeout=600,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0 and "modes ok" in p.stdout, (flip, p.stdout[-2000:], p.stderr[-3000:])


@pytest.mark.parametrize("mode", ["0", "1"])
def test_zstd_decoder_modes(mode):
    """BLOSC_AMD_ZSTD2=0 (one wave per frame for everything: k_zstd_streams) and =1 (two-phase path with the tables in LDS);
    the default, =2 (tables in a global scratch), is what tests/test_gpu_zstd.py runs in-process.  The whole Zstd decode
    suite - reference-written frames, corrupted frames with the oracle's verdict, getitem, mixed batches - in a process of
    its own per mode."""
    env = dict(os.environ)
    env["BLOSC_AMD_ZSTD2"] = mode
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_zstd.py"), "-m", "gpu", "-q", "-x", "--no-header",
                        "-p", "no:cacheprovider"], env=env, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0 and " passed" in p.stdout and "failed" not in p.stdout, (p.stdout[-3000:], p.stderr[-2000:])


# every other environment switch the library reads, one process each: mode_check.py with the entropy-coded formats switched in
# (round trips of every codec under the switch; chunks written here are read by the oracle and by our own decoder)
OTHERS = [{"BLOSC_AMD_PERIODIC": "0"}, {"BLOSC_AMD_CONTEXTS": "1"}, {"BLOSC_AMD_LZ4HC": "0"}, {"BLOSC_AMD_ZSTD_TABLES": "0"}, {"BLOSC_AMD_ZSTD_SEARCH": "1"},
          {"BLOSC_AMD_ZSTD_HUFFMAN": "1", "BLOSC_AMD_ZSTD_TABLES": "1"}, {"BLOSC_AMD_ZLIB_DYNAMIC": "0"}, {"BLOSC_AMD_ZLIB_SEARCH": "0"}, {"BLOSC_AMD_TABLE_CACHE": "0"},
          {"BLOSC_AMD_DEBUG": "1", "BLOSC_AMD_HOSTTIME": "1"}, {"BLOSC_AMD_DEBUG_COST": "1", "BLOSC_AMD_ARENA_SKEW_KIB": "36"}, {"BLOSC_AMD_FUSE": "0", "BLOSC_AMD_SINGLE_QUEUE": "1", "BLOSC_AMD_PERIODIC": "0"}]


@pytest.mark.parametrize("env_extra", OTHERS, ids=["+".join(k[10:] + "=" + v for k, v in e.items()) for e in OTHERS])
def test_other_switches(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    env["BLOSC_MODE_CHECK_Z"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "mode_check.py")], env=env, timeout=900,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0 and "modes ok" in p.stdout, (env_extra, p.stdout[-2000:], p.stderr[-3000:])


def test_every_switch_the_library_reads_is_listed_here():
    """The set of BLOSC_AMD_* names in the product sources = the set this file (and the instrumented build's three dump switches) exercises."""
    import glob
    import re
    names = set()
    for f in glob.glob(os.path.join(ROOT, "c-blosc_amd", "csrc", "*")):
        names |= set(re.findall(r"BLOSC_AMD_[A-Z0-9_]+", open(f, errors="replace").read()))
    covered = set(SWITCHES) | {k for e in OTHERS for k in e} | {"BLOSC_AMD_ZSTD2"}
    profile_build_only = {"BLOSC_AMD_DEC_PROFILE", "BLOSC_AMD_ENC_PROFILE", "BLOSC_AMD_ZSTD_PROFILE"}
    assert names - profile_build_only == covered, (sorted(names - profile_build_only - covered), sorted(covered - names))
