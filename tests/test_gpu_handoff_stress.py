"""GPU: the in-kernel hand-offs under stress.  The decode kernel hands a block from the waves that decoded its streams to
the wave that unshuffles it through one XCD's L2 with a drained store queue + a RELAXED agent-scope counter (k_decode.hip: decode_one_stream),
the encode kernel hands the shuffled block to its encoders the same way (enc_shuffle.h: shuffle_block_task).  That is outside the formal
memory model (the argument is next to the code), so it gets a litmus of its own: two host threads, each with its own context, run their
persistent kernels AT THE SAME TIME on the same XCDs over interleaved mixed batches - byte shuffle typesize 8 / 4 / 2 / 16, bitshuffle, both
codecs, blocks of 32 KiB ... 1 MiB - for >= 2000 launches in total, and every launch's output is compared with the expected bytes."""
import threading

import numpy as np
import pytest

from helpers import DATASETS, orc_compress

pytestmark = pytest.mark.gpu


def test_two_contexts_hammer_the_handoffs(pkg, oracle):
    import torch
    dev = torch.device("cuda:0")
    LAUNCHES = 1000                                   # per thread and direction pair
    specs = [("bench19", 8, 1, "lz4", 0), ("linspace", 8, 1, "lz4", 32768), ("randwalk", 4, 1, "blosclz", 0), ("bench19", 4, 2, "lz4", 65536),
             ("smallints", 2, 1, "lz4", 0), ("bench19", 16, 1, "lz4", 131072), ("zeros", 8, 1, "blosclz", 0), ("arange", 4, 2, "blosclz", 0)]
    errors = []

    def worker(tid):
        try:
            n = (1 << 20) + 4096 * tid
            hosts, chunks = [], []
            for k, (dname, T, shuf, codec, bs) in enumerate(specs):
                d = DATASETS[dname](n + 64 * k)
                r, ch = orc_compress(oracle, d, T, 5, shuf, codec, blocksize=bs)
                assert r > 0
                hosts.append(d); chunks.append(ch)
            nch = len(specs)
            want = [torch.from_numpy(h).to(dev) for h in hosts]
            comp = [torch.from_numpy(c).to(dev) for c in chunks]
            back = [torch.zeros(h.size, dtype=torch.uint8, device=dev) for h in hosts]
            recomp = [torch.zeros(h.size + 16, dtype=torch.uint8, device=dev) for h in hosts]
            bd = pkg.DeviceBatch([c.data_ptr() for c in comp], [c.numel() for c in comp], [b.data_ptr() for b in back], [b.numel() for b in back])
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for it in range(LAUNCHES):
                    for b in back:
                        b.zero_()
                    assert bd.decompress(stream.cuda_stream) == 0
                    assert bd.results() == [h.size for h in hosts], (tid, it, bd.results())
                    for k in range(nch):
                        if not torch.equal(back[k], want[k]):
                            bad = int((back[k] != want[k]).nonzero()[0])
                            raise AssertionError(f"thread {tid} launch {it} chunk {k} ({specs[k]}): first wrong byte at {bad}")
                    if it % 10 == 0:                      # the encode kernel's hand-off as well: compress the same data, decode what it wrote
                        T, shuf, codec = specs[it // 10 % nch][1], specs[it // 10 % nch][2], specs[it // 10 % nch][3]
                        k = it // 10 % nch
                        bc = pkg.DeviceBatch([want[k].data_ptr()], [want[k].numel()], [recomp[k].data_ptr()], [recomp[k].numel()])
                        assert bc.compress(T, 5, shuf, codec.encode(), 0, stream.cuda_stream) == 0
                        cb = bc.results()[0]
                        assert cb > 0
                        b2 = pkg.DeviceBatch([recomp[k].data_ptr()], [cb], [back[k].data_ptr()], [back[k].numel()])
                        back[k].zero_()
                        assert b2.decompress(stream.cuda_stream) == 0 and torch.equal(back[k], want[k]), (tid, it, k)
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
