"""GPU: the C-level multi-GPU entry points (include/blosc_gpu.h: blosc_gpu_partition, blosc_gpu_compress_batch_multi,
blosc_gpu_decompress_batch_multi) - one host thread per device inside one process, contiguous chunk ranges (SURVEY 8e's floor(c G / n)
rule), no exchange between the devices.  This box has ONE GPU: the calls are made with ndev = 2 and 3 over devices {0, 0(, 0)}, which runs
the whole mechanism - partition, worker threads bound to their device, a context per worker, per-chunk results in the caller's arrays - on
the hardware that is there; with more GPUs only the device ids change.  Host buffers and device buffers."""
import ctypes as C

import numpy as np
import pytest

from helpers import DATASETS, orc_decompress

pytestmark = pytest.mark.gpu


def test_partition_rule(lib):
    lo, hi = C.c_size_t(), C.c_size_t()
    for n in (1, 2, 7, 128, 4097):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                assert lib.blosc_gpu_partition(n, w, r, C.byref(lo), C.byref(hi)) == 0
                seen += list(range(lo.value, hi.value))
                for c in range(lo.value, hi.value):
                    assert c * w // n == r
            assert seen == list(range(n))
    assert lib.blosc_gpu_partition(5, 0, 0, C.byref(lo), C.byref(hi)) < 0
    assert lib.blosc_gpu_device_count() >= 1


@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_multi_calls_on_host_buffers(pkg, lib, oracle, ndev):
    n = 11
    datas = [DATASETS[["bench19", "linspace", "randwalk", "zeros"][k % 4]]((1 << 20) + 4096 * k) for k in range(n)]
    outs = [np.zeros(d.size + 16, np.uint8) for d in datas]
    vp, sz = C.c_void_p, C.c_size_t
    src = (vp * n)(*[d.ctypes.data for d in datas]); nb = (sz * n)(*[d.size for d in datas])
    dst = (vp * n)(*[o.ctypes.data for o in outs]); ds = (sz * n)(*[o.size for o in outs]); res = (C.c_int * n)()
    devs = (C.c_int * ndev)(*([0] * ndev))
    assert lib.blosc_gpu_compress_batch_multi(ndev, devs, 5, 1, 8, b"lz4", 0, n, src, nb, dst, ds, res) == 0
    cb = list(res)
    assert all(c > 0 for c in cb)
    for k in range(n):                                                          # every chunk is a stock chunk
        r, back = orc_decompress(oracle, outs[k][:cb[k]], datas[k].size)
        assert r == datas[k].size and np.array_equal(back, datas[k]), k
    backs = [np.zeros(d.size, np.uint8) for d in datas]
    csrc = (vp * n)(*[o.ctypes.data for o in outs]); cs = (sz * n)(*cb)
    bdst = (vp * n)(*[b.ctypes.data for b in backs]); bs = (sz * n)(*[b.size for b in backs]); res2 = (C.c_int * n)()
    assert lib.blosc_gpu_decompress_batch_multi(ndev, devs, n, csrc, cs, bdst, bs, res2) == 0
    assert list(res2) == [d.size for d in datas]
    for k in range(n):
        assert np.array_equal(backs[k], datas[k]), k
    assert lib.blosc_gpu_compress_batch_multi(1, (C.c_int * 1)(99), 5, 1, 8, b"lz4", 0, n, src, nb, dst, ds, res) < 0      # a device this node does not have


def test_multi_calls_on_device_buffers(pkg, lib):
    import torch
    dev = torch.device("cuda:0")
    n, csz = 9, 2 << 20
    hosts = [DATASETS[["bench19", "linspace", "smallints"][k % 3]](csz) for k in range(n)]
    src = torch.stack([torch.from_numpy(h) for h in hosts]).to(dev)
    comp = torch.zeros((n, csz + 16), dtype=torch.uint8, device=dev); back = torch.zeros((n, csz), dtype=torch.uint8, device=dev)
    vp, sz = C.c_void_p, C.c_size_t
    res = (C.c_int * n)()
    assert lib.blosc_gpu_compress_batch_multi(2, None, 5, 1, 4, b"lz4", 0, n, (vp * n)(*[src[i].data_ptr() for i in range(n)]), (sz * n)(*([csz] * n)),
                                              (vp * n)(*[comp[i].data_ptr() for i in range(n)]), (sz * n)(*([csz + 16] * n)), res) in (0, -1)
    # devices NULL = 0 .. ndev-1: device 1 does not exist on a one-GPU box -> refused as a whole (-1), nothing half done; with {0, 0} it runs
    devs = (C.c_int * 2)(0, 0)
    assert lib.blosc_gpu_compress_batch_multi(2, devs, 5, 1, 4, b"lz4", 0, n, (vp * n)(*[src[i].data_ptr() for i in range(n)]), (sz * n)(*([csz] * n)),
                                              (vp * n)(*[comp[i].data_ptr() for i in range(n)]), (sz * n)(*([csz + 16] * n)), res) == 0
    cb = list(res)
    assert all(c > 0 for c in cb)
    res2 = (C.c_int * n)()
    assert lib.blosc_gpu_decompress_batch_multi(2, devs, n, (vp * n)(*[comp[i].data_ptr() for i in range(n)]), (sz * n)(*cb),
                                                (vp * n)(*[back[i].data_ptr() for i in range(n)]), (sz * n)(*([csz] * n)), res2) == 0
    torch.cuda.synchronize()
    assert list(res2) == [csz] * n and torch.equal(back, src)


def test_plain_call_after_multi_call_stays_on_the_callers_device(pkg, lib):
    """after a _multi call the library's contexts live on devices 0 .. N-1 (which worker got which context is a race).  A plain
    call that never named a device must run on the caller's CURRENT device - not on whatever device the first free context landed on - and must
    leave the calling thread's HIP device alone.  With one GPU the device lists alias device 0 (the selection code still runs); with two or
    more the workers really spread, the order {last, ..., 0} puts context 0 on the LAST device, and the plain call's pointers are device-0 memory."""
    import torch
    ngpu = lib.blosc_gpu_device_count()
    order = list(range(ngpu - 1, -1, -1)) if ngpu > 1 else [0, 0]
    nd = len(order)
    n, csz = 2 * nd + 1, 1 << 20
    vp, sz = C.c_void_p, C.c_size_t
    hosts = [DATASETS["bench19"](csz) for _ in range(n)]
    outs = [np.zeros(csz + 16, np.uint8) for _ in range(n)]
    res = (C.c_int * n)()
    assert lib.blosc_gpu_compress_batch_multi(nd, (C.c_int * nd)(*order), 5, 1, 8, b"lz4", 0, n, (vp * n)(*[h.ctypes.data for h in hosts]), (sz * n)(*([csz] * n)),
                                              (vp * n)(*[o.ctypes.data for o in outs]), (sz * n)(*([csz + 16] * n)), res) == 0
    assert all(c > 0 for c in res)
    torch.cuda.set_device(0)
    dev0 = torch.device("cuda:0")
    src = torch.from_numpy(hosts[0]).to(dev0)
    comp = torch.zeros(csz + 16, dtype=torch.uint8, device=dev0); back = torch.zeros(csz, dtype=torch.uint8, device=dev0)
    for _ in range(4):                                   # several calls: every free context gets its turn at being "the first free one"
        b = pkg.DeviceBatch([src.data_ptr()], [csz], [comp.data_ptr()], [csz + 16])
        assert b.compress(8, 5, 1, b"lz4") == 0
        cb = b.results()[0]
        assert cb > 0
        d = pkg.DeviceBatch([comp.data_ptr()], [cb], [back.data_ptr()], [csz])
        assert d.decompress() == 0 and d.results() == [csz]
        torch.cuda.synchronize()
        assert torch.equal(back, src)
        assert torch.cuda.current_device() == 0
