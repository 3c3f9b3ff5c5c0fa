"""CPU: the byte (un)shuffle that the persistent encode / decode kernels run as work of their own waves (k_encode.hip:
shuffle_block_wave_T, shuffle_block_wave_detect with the periodic-plane shortcut and emit_periodic_stream; k_decode.hip:
unshuffle_block_wave) - the device source on the wavefront emulator, against the oracle's shuffle (pinned to the reference's
blosc/shuffle-generic.h) and the oracle's stream decoders."""
import ctypes as C

import os
import numpy as np
import pytest

from helpers import DATASETS, ptr
from test_wave_emu_encoders import emu  # noqa: F401


SOAK = 7919 * int(os.environ.get("BLOSC_EMU_SEED", "0"))      # soak runs (BLOSC_EMU_SEED=1, 2, ...): every random draw of this file moves

def _api(emu):
    emu.emu_shuffle_block.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
    emu.emu_shuffle_block.restype = C.c_uint
    emu.emu_periodic_stream.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_uint, C.c_int]
    emu.emu_periodic_stream.restype = C.c_uint
    return emu


@pytest.mark.parametrize("T", [2, 4, 8, 16, 3, 5, 6, 7, 12, 17, 24, 25, 31, 32])      # 2 / 4 / 8 / 16: register transposes; the others: the LDS-tile forms of round 4
def test_shuffle_and_unshuffle_equal_the_oracles(emu, oracle, T):
    e = _api(emu)
    rng = np.random.default_rng(T + SOAK)
    for bsize in [256 * T, 1024 * T, 1280 * T, 4096 * T, 256 * 1024] + ([128 * T, 384 * T + 7] if T not in (2, 4, 8, 16) else []):
        for data in (rng.integers(0, 256, bsize, dtype=np.uint8), DATASETS["bench19"](bsize), DATASETS["linspace"](bsize)):
            want = np.zeros(bsize, np.uint8)
            oracle.orc_shuffle(T, bsize, ptr(data), ptr(want))
            got = np.full(bsize + 64, 0xEE, np.uint8)
            if T in (2, 4, 8, 16) and (bsize // T) % 256:      # (the register forms of the fused SHUFFLE are only ever given whole rows + the generic tail; sizes with a ragged tail go through mode 2 below)
                continue
            e.emu_shuffle_block(T, 0, ptr(data), ptr(got), bsize, None)
            assert np.array_equal(got[:bsize], want) and np.all(got[bsize:] == 0xEE), (T, bsize)
            back = np.full(bsize + 64, 0xEE, np.uint8)
            e.emu_shuffle_block(T, 2, ptr(want), ptr(back), bsize, None)
            assert np.array_equal(back[:bsize], data) and np.all(back[bsize:] == 0xEE), (T, bsize)
    # unshuffle: blocks that are not a multiple of 256 elements, and leftover bytes behind the last whole element
    for bsize in [T * 100, T * 300 + 3, T * 1030 + T - 1, 4096 * T + 5]:
        data = rng.integers(0, 256, bsize, dtype=np.uint8)
        sh = np.zeros(bsize, np.uint8); oracle.orc_shuffle(T, bsize, ptr(data), ptr(sh))
        back = np.full(bsize + 64, 0xEE, np.uint8)
        e.emu_shuffle_block(T, 2, ptr(sh), ptr(back), bsize, None)
        assert np.array_equal(back[:bsize], data) and np.all(back[bsize:] == 0xEE), (T, bsize)


@pytest.mark.parametrize("T", [2, 4, 8, 16])
def test_periodic_planes_are_found_and_written_as_one_match(emu, oracle, T):
    """A plane whose 256-byte rows are all equal is not written by the shuffle (only its first row is) and not searched by the
    match finder: its stream is `period` literals + one match.  Planes of every period that divides 256, mixed with planes that
    are not periodic, or periodic only up to some row."""
    e = _api(emu)
    oracle.orc_blosclz_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(40 + T + SOAK)
    N = 8192                                                                  # bytes per plane
    for trial in range(12):
        planes = []; expect = []
        for j in range(T):
            kind = (trial + j) % 4
            if kind == 0:
                p = int(rng.choice([1, 2, 4, 8, 16, 32, 64, 128, 256]))
                planes.append(np.resize(rng.integers(0, 256, p, dtype=np.uint8), N)); expect.append(p)
            elif kind == 1:
                planes.append(rng.integers(0, 256, N, dtype=np.uint8)); expect.append(0)
            elif kind == 2:                                                   # periodic, broken somewhere behind the first row
                a = np.resize(rng.integers(0, 256, 64, dtype=np.uint8), N).copy(); a[int(rng.integers(256, N))] ^= 0x5A
                planes.append(a); expect.append(0)
            else:
                planes.append(np.zeros(N, np.uint8)); expect.append(1)
        # true period of a periodic plane may be smaller than the one it was built with
        data = np.ascontiguousarray(np.stack(planes, 1)).reshape(-1)          # element-major block
        got = np.zeros(N * T, np.uint8)
        per = (C.c_uint * 16)()
        mask = e.emu_shuffle_block(T, 1, ptr(data), ptr(got), N * T, per)
        for j in range(T):
            plane = got[j * N:(j + 1) * N]
            if expect[j] == 0:
                assert not (mask >> j) & 1 and per[j] == 0 and np.array_equal(plane, planes[j]), (trial, j)
                continue
            assert (mask >> j) & 1 and 1 <= per[j] <= expect[j] and expect[j] % per[j] == 0, (trial, j, per[j], expect[j])
            assert np.array_equal(plane[:256], planes[j][:256])
            assert np.array_equal(np.resize(planes[j][:per[j]], N), planes[j])
            for lz4 in (1, 0):
                out = np.full(600, 0xEE, np.uint8)
                r = e.emu_periodic_stream(ptr(np.ascontiguousarray(plane[:256])), N, ptr(out), 512, per[j], lz4)
                assert 0 < r < 340 and np.all(out[512:] == 0xEE)
                back = np.zeros(N + 8, np.uint8)
                f = oracle.orc_lz4_decompress if lz4 else oracle.orc_blosclz_decompress
                assert f(ptr(out), r, ptr(back), N) == N and np.array_equal(back[:N], planes[j]), (trial, j, lz4)


@pytest.mark.parametrize("fmt,mode", [(1, 0), (0, 0), (1, 3)], ids=["lz4", "blosclz", "lz4hc"])
@pytest.mark.parametrize("T", [2, 4, 8, 16])
def test_block_through_the_kernels_task_functions(emu, oracle, T, fmt, mode):
    """shuffle_block_task (periodic planes noted in their stream descriptors) followed by encode_one_stream for every plane - the two
    task kinds of the persistent encode kernel, as it calls them: every plane's stream must decode to that plane (constant and
    short-period planes through emit_periodic_stream, the others through the match finder), incompressible planes report 0."""
    emu.emu_encode_block.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.POINTER(C.c_int)]
    oracle.orc_blosclz_decompress.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    rng = np.random.default_rng(50 + T + fmt + SOAK)
    ne = 8192
    for trial in range(4):
        planes = []
        for j in range(T):
            kind = (trial + j) % 5
            if kind == 0: planes.append(np.zeros(ne, np.uint8))
            elif kind == 1: planes.append(np.resize(rng.integers(0, 256, int(rng.choice([2, 16, 128, 256])), dtype=np.uint8), ne))
            elif kind == 2: planes.append(rng.integers(0, 256, ne, dtype=np.uint8))
            elif kind == 3: planes.append(rng.integers(0, 3, ne, dtype=np.uint8))
            else: planes.append(np.resize(rng.integers(0, 256, 700, dtype=np.uint8), ne))
        data = np.ascontiguousarray(np.stack(planes, 1)).reshape(-1)
        slot = ne + 64
        out = np.full(T * slot + 64, 0xEE, np.uint8)
        res = (C.c_int * 16)()
        emu.emu_encode_block(T, fmt, mode, 9 if mode == 3 else 5, ptr(data), ne * T, ptr(out), slot, res)
        assert np.all(out[T * slot:] == 0xEE)
        for j in range(T):
            r = res[j]
            kind = (trial + j) % 5
            if kind == 2:
                assert r == 0
                continue
            assert 0 < r < ne, (trial, j, r)
            if kind in (0, 1):
                assert r < 340                                              # the periodic shortcut: a few literals and one match
            s = out[j * slot:j * slot + r].copy()
            back = np.zeros(ne + 8, np.uint8)
            f = oracle.orc_lz4_decompress if fmt == 1 else oracle.orc_blosclz_decompress
            assert f(ptr(s), r, ptr(back), ne) == ne and np.array_equal(back[:ne], planes[j]), (trial, j)
