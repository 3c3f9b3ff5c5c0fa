"""GPU: one GPU's share of BASELINE.json's configuration #5 - 512 chunks of 64 MiB = 32 GiB, 262 144 streams, arenas beyond 4 GiB
(SURVEY 8d cfg #5: 256 GiB over 8 GPUs; /root/reference/blosc/blosc.h:40 BLOSC_MAX_BUFFERSIZE bounds a chunk, not a batch) - as a parity
test: every byte of the batch compared in both directions, the batch made of DIFFERENT chunks (four data classes taking turns, so the
queues deal planes of different cost side by side), reference-written chunks decoded here, and the first, a middle and the last chunk
written here read by the reference.  What this size is for: 32-bit offsets, queue lengths and ticket counters that 128 chunks never reach
(round 4 found one such limit: tests/test_gpu_spans.py::test_self_span_base_beyond_24_bits_is_not_taken)."""
import numpy as np
import pytest

from helpers import DATASETS, ref_compress, ref_decompress

pytestmark = pytest.mark.gpu
CSZ = 64 << 20
NCHUNKS = 512
KINDS = ["bench19", "linspace", "randwalk", "arange"]


def test_512_chunks_of_64_mib_both_directions(pkg, lib, ref):
    import torch
    if ref is None:
        pytest.skip("needs the reference (oracle/_ref) as writer and reader")
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip(f"needs about 200 GiB of device memory for three 32 GiB buffers and the arenas, {free >> 30} GiB free")
    T, clevel, shuffle, codec = 8, 5, 1, b"lz4"
    hosts = [DATASETS[k](CSZ) for k in KINDS]
    stocks = []
    for h in hosts:
        r, st = ref_compress(ref, h, T, clevel, shuffle, codec, nthreads=8)
        assert r > 0
        stocks.append(st)
    nk = len(KINDS)
    src = torch.empty((NCHUNKS, CSZ), dtype=torch.uint8, device=dev)
    comp = torch.zeros((NCHUNKS, CSZ + 256), dtype=torch.uint8, device=dev)
    back = torch.empty((NCHUNKS, CSZ), dtype=torch.uint8, device=dev)
    for k, (h, st) in enumerate(zip(hosts, stocks)):
        rows = len(range(k, NCHUNKS, nk))
        src[k::nk].copy_(torch.from_numpy(h).to(dev).unsqueeze(0).expand(rows, CSZ))
        comp[k::nk, :st.size].copy_(torch.from_numpy(st).to(dev).unsqueeze(0).expand(rows, st.size))
    bd = pkg.DeviceBatch([comp[i].data_ptr() for i in range(NCHUNKS)], [CSZ + 16] * NCHUNKS,
                         [back[i].data_ptr() for i in range(NCHUNKS)], [CSZ] * NCHUNKS)
    bc = pkg.DeviceBatch([src[i].data_ptr() for i in range(NCHUNKS)], [CSZ] * NCHUNKS,
                         [comp[i].data_ptr() for i in range(NCHUNKS)], [CSZ + 16] * NCHUNKS)

    def same_as_source(what):
        for lo in range(0, NCHUNKS, 64):      # (64 chunks at a time: torch.equal on 32 GiB at once needs no temporary, but a failure should name the place)
            assert torch.equal(back[lo:lo + 64], src[lo:lo + 64]), (what, "chunks", lo, lo + 64)

    # ---- reference-written chunks -> here, twice (the second call runs with the first one's queue order) ----
    for rep in range(2):
        back.zero_()
        assert bd.decompress() == 0
        assert bd.results() == [CSZ] * NCHUNKS
        same_as_source(("decompress of reference-written chunks", rep))
    # ---- here -> the reference (ends and middle), and back through our own decoder (every byte) ----
    for rep in range(2):
        comp.zero_()
        assert bc.compress(T, clevel, shuffle, codec) == 0
        cb = bc.results()
        assert len(cb) == NCHUNKS and all(0 < c <= CSZ + 16 for c in cb)
        for i in (0, 1, 2, 3, NCHUNKS // 2 + 1, NCHUNKS - 2, NCHUNKS - 1):
            rr, out = ref_decompress(ref, comp[i][:cb[i]].cpu().numpy(), CSZ)
            assert rr == CSZ and np.array_equal(out, hosts[i % nk]), ("the reference cannot read chunk", i, rep)
        # chunks of one data class are written identically wherever they sit in the batch
        for k in range(nk):
            assert len({cb[i] for i in range(k, NCHUNKS, nk)}) == 1, ("cbytes differ inside one data class", KINDS[k])
            first = comp[k][:cb[k]]
            for i in (k + nk * 37, k + nk * 127):
                assert torch.equal(comp[i][:cb[k]], first), ("chunk bytes differ inside one data class", KINDS[k], i)
        back.zero_()
        assert bd.decompress() == 0 and bd.results() == [CSZ] * NCHUNKS
        same_as_source(("round trip", rep))
