"""blpk.py — a many-chunk container in Bloscpack's file layout, streamed through the BATCHED GPU calls.

c-blosc compresses one buffer of at most 2 GiB per call; files are handled by callers such as Bloscpack (`README.md:173-177`
of the reference points there), which cut the data into chunks, run `blosc_compress` on each and store the chunks behind a
small header with an offset table.  This module is that caller for libblosc_amd: the chunks of a file go through
`blosc_gpu_compress_batch_host` / `blosc_gpu_decompress_batch_host` several hundred at a time instead of one call per chunk, so a file
is a handful of launches.

File layout (Bloscpack format version 3, written from its published format description; **parity unpinned**: Bloscpack is
a separate project, neither in the reference tree nor in this image, so no byte-for-byte comparison with its own files was
possible - what IS pinned: every chunk inside is an ordinary c-blosc chunk, read by the reference and the oracle in the tests):

    bytes 0-3   magic "blpk"          4  format version (3)        5  options (bit 0: offsets, bit 1: metadata)
    6  checksum (0 none, 1 adler32, 2 crc32)      7  typesize      8-11  chunk_size (int32)    12-15  last_chunk (int32)
    16-23  nchunks (int64)            24-31  max_app_chunks (int64)
    [offsets: (nchunks + max_app_chunks) x int64, from the start of the file, -1 = unused]
    chunk 0 [+ 4-byte checksum of the compressed chunk, little endian], chunk 1 ...

Only what the path needs: no metadata section is written (a file that has one is read past it), checksums none / adler32 /
crc32.  There is no CPU implementation here: without the library and a GPU the calls fail.
"""
import ctypes as C
import os
import struct
import zlib

import numpy as np

MAGIC = b"blpk"
FORMAT_VERSION = 3
HEADER_LENGTH = 32
METADATA_HEADER_LENGTH = 32
CHECKSUMS = ("None", "adler32", "crc32")
_OPT_OFFSETS, _OPT_METADATA = 1, 2


class BlpkError(ValueError):
    pass


def _digest(kind, data):
    if kind == 1:
        return struct.pack("<I", zlib.adler32(data) & 0xffffffff)
    if kind == 2:
        return struct.pack("<I", zlib.crc32(data) & 0xffffffff)
    return b""


def pack_header(nchunks, chunk_size, last_chunk, typesize, checksum=1, offsets=True, max_app_chunks=0):
    if not 0 <= checksum < len(CHECKSUMS):
        raise BlpkError("unknown checksum")
    return MAGIC + struct.pack("<BBBBiiqq", FORMAT_VERSION, _OPT_OFFSETS if offsets else 0, checksum, typesize & 0xff,
                               chunk_size, last_chunk, nchunks, max_app_chunks)


def unpack_header(buf):
    if len(buf) < HEADER_LENGTH:
        raise BlpkError("file shorter than a bloscpack header")
    if buf[:4] != MAGIC:
        raise BlpkError("not a bloscpack file (magic)")
    version, options, checksum, typesize, chunk_size, last_chunk, nchunks, max_app = struct.unpack("<BBBBiiqq", buf[4:HEADER_LENGTH])
    if version != FORMAT_VERSION:
        raise BlpkError(f"format version {version} (this reader knows {FORMAT_VERSION})")
    if checksum >= len(CHECKSUMS):
        raise BlpkError(f"checksum kind {checksum} not supported")
    if nchunks < 0 or max_app < 0 or chunk_size == 0:
        raise BlpkError("header needs nchunks and chunk_size")
    return dict(options=options, checksum=checksum, typesize=typesize, chunk_size=chunk_size, last_chunk=last_chunk,
                nchunks=nchunks, max_app_chunks=max_app, offsets=bool(options & _OPT_OFFSETS), metadata=bool(options & _OPT_METADATA))


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def pack(lib, data, fh, chunk_size=1 << 20, typesize=8, clevel=5, shuffle=1, cname=b"lz4", checksum=1, batch_bytes=1 << 30):
    """Compress `data` (numpy, any dtype) into the open binary file `fh`.  Chunks go to the GPU `batch_bytes` at a time
    through blosc_gpu_compress_batch_host.  Returns (nchunks, bytes written)."""
    a = np.ascontiguousarray(data).view(np.uint8).ravel()
    n = a.size
    if chunk_size <= 0 or chunk_size > (1 << 31) - 17:
        raise BlpkError("chunk_size out of range")
    nchunks = (n + chunk_size - 1) // chunk_size if n else 0
    last = n - (nchunks - 1) * chunk_size if nchunks else 0
    start = fh.tell()
    fh.write(pack_header(nchunks, chunk_size, last, typesize, checksum))
    off_pos = fh.tell()
    offsets = np.full(nchunks, -1, "<i8")
    fh.write(offsets.tobytes())
    per_batch = max(1, batch_bytes // chunk_size)
    for b0 in range(0, nchunks, per_batch):
        b1 = min(nchunks, b0 + per_batch)
        srcs = [a[k * chunk_size:min(n, (k + 1) * chunk_size)] for k in range(b0, b1)]
        dsts = [np.empty(s.size + 16, np.uint8) for s in srcs]
        m = b1 - b0
        ssz = (C.c_size_t * m)(*[s.size for s in srcs]); dsz = (C.c_size_t * m)(*[d.size for d in dsts])
        res = (C.c_int * m)()
        rc = lib.blosc_gpu_compress_batch_host(clevel, shuffle, typesize, cname, 0, m, _ptr_array(srcs), ssz, _ptr_array(dsts), dsz, res)
        if rc != 0 or any(r <= 0 for r in res):
            raise BlpkError(f"compression failed (rc {rc}, results {list(res)[:4]}...)")
        for k, (d, r) in enumerate(zip(dsts, res)):
            offsets[b0 + k] = fh.tell() - start
            chunk = d[:r].tobytes()
            fh.write(chunk); fh.write(_digest(checksum, chunk))
    end = fh.tell()
    fh.seek(off_pos); fh.write(offsets.tobytes()); fh.seek(end)
    return nchunks, end - start


def unpack(lib, fh, batch_bytes=1 << 30, verify=True):
    """Read a bloscpack file from the open binary file `fh`; returns the plain bytes as a numpy uint8 array.  Chunks go to the
    GPU `batch_bytes` at a time through blosc_gpu_decompress_batch_host."""
    blob = fh.read()
    h = unpack_header(blob)
    pos = HEADER_LENGTH
    if h["metadata"]:                                  # skip: magic_format(8) options checksum codec level meta_size max_meta_size meta_comp_size user_codec(8)
        if len(blob) < pos + METADATA_HEADER_LENGTH:
            raise BlpkError("truncated metadata header")
        meta_checksum = blob[pos + 9]
        max_meta = struct.unpack("<I", blob[pos + 16:pos + 20])[0]
        dlen = {0: 0, 1: 4, 2: 4, 3: 16, 4: 20, 5: 28, 6: 32, 7: 48, 8: 64}.get(meta_checksum)
        if dlen is None:
            raise BlpkError("metadata checksum kind")
        pos += METADATA_HEADER_LENGTH + max_meta + dlen
    nch = h["nchunks"]
    dlen = 4 if h["checksum"] else 0
    if h["offsets"]:
        tot = nch + h["max_app_chunks"]
        if len(blob) < pos + 8 * tot:
            raise BlpkError("truncated offset table")
        offs = np.frombuffer(blob, "<i8", tot, pos)[:nch].astype(np.int64)
        pos += 8 * tot
    else:
        offs = None
    chunks = []; sizes = []
    for k in range(nch):
        o = int(offs[k]) if offs is not None else pos
        if o < 0 or o + 16 > len(blob):
            raise BlpkError(f"chunk {k}: offset outside the file")
        nbytes, _bs, cbytes = struct.unpack("<iii", blob[o + 4:o + 16])
        if cbytes < 16 or o + cbytes + dlen > len(blob) or nbytes < 0:
            raise BlpkError(f"chunk {k}: header sizes outside the file")
        c = np.frombuffer(blob, np.uint8, cbytes, o)
        if verify and dlen and _digest(h["checksum"], c.tobytes()) != blob[o + cbytes:o + cbytes + dlen]:
            raise BlpkError(f"chunk {k}: checksum mismatch")
        chunks.append(c); sizes.append(nbytes)
        pos = o + cbytes + dlen
    total = int(sum(sizes))
    out = np.empty(total, np.uint8)
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    k0 = 0
    while k0 < nch:
        k1 = k0; acc = 0
        while k1 < nch and (k1 == k0 or acc + sizes[k1] <= batch_bytes):
            acc += sizes[k1]; k1 += 1
        m = k1 - k0
        srcs = [np.ascontiguousarray(chunks[k]) for k in range(k0, k1)]
        dsts = [out[starts[k]:starts[k + 1]] for k in range(k0, k1)]
        ssz = (C.c_size_t * m)(*[s.size for s in srcs]); dsz = (C.c_size_t * m)(*[d.size for d in dsts])
        res = (C.c_int * m)()
        rc = lib.blosc_gpu_decompress_batch_host(m, _ptr_array(srcs), ssz, _ptr_array(dsts), dsz, res)
        if rc != 0 or any(r != d.size for r, d in zip(res, dsts)):
            raise BlpkError(f"decompression failed (rc {rc}, results {list(res)[:4]}...)")
        k0 = k1
    return out


def pack_file(lib, src_path, dst_path, **kw):
    data = np.fromfile(src_path, np.uint8)
    with open(dst_path, "wb") as fh:
        return pack(lib, data, fh, **kw)


def unpack_file(lib, src_path, dst_path=None, **kw):
    with open(src_path, "rb") as fh:
        out = unpack(lib, fh, **kw)
    if dst_path is not None:
        out.tofile(dst_path)
    return out
