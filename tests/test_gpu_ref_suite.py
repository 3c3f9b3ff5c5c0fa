"""GPU: the reference's OWN acceptance programs against the drop-in (SURVEY §2 "acceptance suite to re-run
against the drop-in .so", §8b, §8f-1).

tests/ref_suite/Makefile compiles /root/reference/tests/test_*.c, compat/filegen.c, bench/bench.c and
tests/fuzz/{fuzz_decompress,fuzz_compress,standalone}.c UNMODIFIED against include/blosc.h and links them with
libblosc_amd.so (SONAME libblosc.so.1).  The binaries are built where /root/reference exists and travel to the GPU
box like oracle/_ref; here they are only executed.  Deviations from a stock build are listed in INTEGRATION.md §7
and asserted below (so a change in behaviour shows up either way).
"""
import glob
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "ref_suite", "_bin")
COMPAT = os.path.join(ROOT, "tests", "golden", "compat")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(tmp_path_factory):
    if not os.path.exists(os.path.join(BIN, "test_api")):
        if os.path.isdir("/root/reference"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "ref_suite")], stdout=subprocess.DEVNULL)
        else:
            pytest.skip("tests/ref_suite/_bin is missing: the binaries are built where /root/reference exists (`make -C tests/ref_suite`, "
                        "also done by __graft_entry__.build()) and travel to the GPU box with the tree")
    # the binaries ask for libblosc.so.1 (the stock SONAME): serve the drop-in under that name
    libdir = tmp_path_factory.mktemp("libblosc")
    os.symlink(os.path.join(ROOT, "c-blosc_amd", "libblosc_amd.so"), os.path.join(libdir, "libblosc.so.1"))
    e = dict(os.environ)
    e["LD_LIBRARY_PATH"] = f"{libdir}:/opt/rocm/lib:" + e.get("LD_LIBRARY_PATH", "")
    return e


def run(env, args, timeout=600, cwd=None):
    # never the repo root: some of the reference's programs (test_bitshuffle_leftovers.c:24,65) write files into their cwd
    p = subprocess.run([os.path.join(BIN, args[0])] + list(args[1:]), env=env, cwd=cwd or tempfile.gettempdir(), timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, errors="replace")
    return p.returncode, p.stdout, p.stderr


@pytest.mark.parametrize("prog", ["test_api", "test_maxout", "test_compressor", "test_nolock", "test_noinit", "test_nthreads"])
def test_minunit_program(env, prog):
    rc, out, err = run(env, [prog])
    assert rc == 0 and "ALL TESTS PASSED" in out, (rc, out[-2000:], err[-2000:])


def test_bitshuffle_leftovers(env, tmp_path):
    """tests/test_bitshuffle_leftovers.c: 641091-byte buffers (not a multiple of 8 elements) through bitshuffle with
    both codecs; the program prints one 'Successful roundtrip!' per case and returns 0."""
    rc, out, err = run(env, ["test_bitshuffle_leftovers"], cwd=tmp_path)      # the program writes two .cdata files into its cwd
    assert rc == 0 and out.count("Successful roundtrip!") == 2 and "error" not in out.lower(), (rc, out[-2000:], err[-2000:])


@pytest.mark.parametrize("prog", ["test_compress_roundtrip", "test_getitem", "test_shuffle_roundtrip_generic", "test_shuffle_roundtrip_sse2", "test_shuffle_roundtrip_avx2"])
def test_csv_program(env, prog):
    """Every row of the reference's parameter list (tests/CMakeLists.txt:66-100 turns each into a ctest case)."""
    stride = os.environ.get("REF_SUITE_STRIDE", "1")
    rc, out, err = run(env, [prog, "--csv", os.path.join(BIN, prog + ".csv"), stride], timeout=1800)
    assert rc == 0 and " 0 failed" in out, (rc, out[-1500:], err[-3000:])


def test_forksafe_deviation(env):
    """tests/test_forksafe.c forks AFTER the parent has compressed and decompresses in the child.  A HIP context does
    not survive fork(), so the child cannot use the parent's device state; the library's atfork handler makes the
    child fail loudly (-1) instead of hanging - never a silent wrong answer.  Listed in INTEGRATION.md §7."""
    rc, out, err = run(env, ["test_forksafe"], timeout=120)
    assert "Child deadlocked" not in out, (out, err)
    assert rc in (0, 1)


@pytest.mark.parametrize("fname", sorted(os.path.basename(f) for f in glob.glob(os.path.join(COMPAT, "*.cdata"))))
def test_filegen_decodes_compat_vector(env, fname):
    """compat/filegen.c:59-104 in decompress mode over the golden vectors (what compat/CMakeLists.txt runs)."""
    rc, out, err = run(env, ["filegen", "decompress", os.path.join(COMPAT, fname)])
    if "snappy" in fname:
        assert rc != 0 and "Decompression error" in out      # codec this build does not carry: -5 like a stock build without it
    else:
        assert rc == 0 and "Decompression successful!" in out, (rc, out, err)


@pytest.mark.parametrize("codec,filt,T", [("lz4", "shuffle", 8), ("blosclz", "shuffle", 8), ("lz4", "bitshuffle", 4), ("lz4", "noshuffle", 4)])
def test_bench_single(env, codec, filt, T):
    """bench/bench.c 'single' suite (the program BASELINE.json's configs[0] names) on an 8 MiB buffer: every clevel
    round-trips ("OK" per level, bench.c:300-318)."""
    rc, out, err = run(env, ["bench", codec, filt, "single", "1", str(8 << 20), str(T), "19"], timeout=900)
    assert rc == 0 and "FAILED" not in out and "do not match" not in out, (out[-3000:], err[-2000:])
    assert out.count("OK\n") >= 10, out[-3000:]


def test_fuzz_decompress_corpus(env, tmp_path):
    """tests/fuzz/fuzz_decompress.c:14-35 + standalone.c over the compat vectors and seeded mutations of them (bit
    flips, truncations with a patched cbytes field, header-field edits): the harness must survive every input."""
    rng = np.random.default_rng(20240924)
    files = []
    for f in sorted(glob.glob(os.path.join(COMPAT, "*.cdata"))):
        base = np.fromfile(f, np.uint8)
        name = os.path.basename(f)
        shutil.copy(f, tmp_path / name)
        files.append(str(tmp_path / name))
        for k in range(6):
            m = base.copy()
            kind = k % 3
            if kind == 0:      # bit flips in the payload
                for pos in rng.integers(16, m.size, 8):
                    m[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:    # truncate and patch cbytes so that the harness lets it through
                m = m[: int(rng.integers(20, m.size))].copy()
                m[12:16] = np.array([m.size], "<u4").view(np.uint8)
            else:              # header / bstarts edits
                pos = int(rng.integers(2, min(m.size, 16 + 64)))
                if 12 <= pos < 16:
                    pos = 8
                m[pos] = int(rng.integers(0, 256))
            p = tmp_path / f"{name}.mut{k}"
            m.tofile(p)
            files.append(str(p))
    rc, out, err = run(env, ["fuzz_decompress"] + files, timeout=900)
    assert rc == 0, (rc, err[-3000:])
    assert err.count("Done:") == len(files)


def test_fuzz_compress_harness(env, tmp_path):
    """tests/fuzz/fuzz_compress.c: header bytes of the input select clevel / shuffle / typesize / codec, the rest is data."""
    rng = np.random.default_rng(7)
    files = []
    for k in range(24):
        n = int(rng.integers(40, 300000))
        a = rng.integers(0, 4 if k % 2 else 256, n).astype(np.uint8)
        p = tmp_path / f"in{k}"
        a.tofile(p)
        files.append(str(p))
    rc, out, err = run(env, ["fuzz_compress"] + files, timeout=900)
    assert rc == 0, (rc, err[-3000:])
