"""c-blosc_amd — Python-side loader for libblosc_amd.so (ctypes; test / bench plumbing only).

The product is the C-ABI shared library built from ``csrc/`` (see ``include/blosc.h`` and
``include/blosc_gpu.h``).  This module only locates it, declares argument types and offers small
numpy conveniences that mirror how the reference is driven from Python through ctypes
(SURVEY.md §A.8).  There is no CPU implementation here: if the library is missing, ``load()``
raises; if there is no GPU, the library's calls return errors.

The directory name contains a hyphen (it mirrors the reference's repository name), so import it
with ``importlib`` — ``tests/conftest.py`` and ``__graft_entry__.py`` show how.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BLOSC_AMD_LIB", os.path.join(_HERE, "libblosc_amd.so"))   # override: tuning experiments only

STOCK_SYMBOLS = [
    "blosc_init", "blosc_destroy", "blosc_compress", "blosc_compress_ctx", "blosc_decompress",
    "blosc_decompress_ctx", "blosc_getitem", "blosc_get_nthreads", "blosc_set_nthreads",
    "blosc_get_compressor", "blosc_set_compressor", "blosc_compcode_to_compname",
    "blosc_compname_to_compcode", "blosc_list_compressors", "blosc_get_version_string",
    "blosc_get_complib_info", "blosc_free_resources", "blosc_cbuffer_sizes", "blosc_cbuffer_validate",
    "blosc_cbuffer_metainfo", "blosc_cbuffer_versions", "blosc_cbuffer_complib", "blosc_get_blocksize",
    "blosc_set_blocksize", "blosc_set_splitmode",
]
GPU_SYMBOLS = [
    "blosc_gpu_set_device", "blosc_gpu_compress_batch", "blosc_gpu_decompress_batch", "blosc_gpu_getitem",
    "blosc_gpu_compress_batch_host", "blosc_gpu_decompress_batch_host",
    "blosc_gpu_device_count", "blosc_gpu_partition", "blosc_gpu_compress_batch_multi", "blosc_gpu_decompress_batch_multi",
    "blosc_gpu_profile", "blosc_gpu_profile_reset", "blosc_gpu_profile_get",
]

_lib = None


def load():
    """Return the ctypes handle of libblosc_amd.so with argtypes/restypes declared."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C {_HERE}` (hipcc, gfx950). "
            "There is no fallback implementation.")
    L = C.CDLL(LIB_PATH)
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    L.blosc_compress.argtypes = [i, i, sz, sz, vp, vp, sz]
    L.blosc_compress_ctx.argtypes = [i, i, sz, sz, vp, vp, sz, C.c_char_p, sz, i]
    L.blosc_decompress.argtypes = [vp, vp, sz]
    L.blosc_decompress_ctx.argtypes = [vp, vp, sz, i]
    L.blosc_getitem.argtypes = [vp, i, i, vp]
    L.blosc_set_compressor.argtypes = [C.c_char_p]
    L.blosc_get_compressor.restype = C.c_char_p
    L.blosc_list_compressors.restype = C.c_char_p
    L.blosc_get_version_string.restype = C.c_char_p
    L.blosc_cbuffer_complib.restype = C.c_char_p
    L.blosc_cbuffer_complib.argtypes = [vp]
    L.blosc_compname_to_compcode.argtypes = [C.c_char_p]
    L.blosc_compcode_to_compname.argtypes = [i, C.POINTER(C.c_char_p)]
    L.blosc_get_complib_info.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
    L.blosc_cbuffer_sizes.argtypes = [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.blosc_cbuffer_sizes.restype = None
    L.blosc_cbuffer_validate.argtypes = [vp, sz, C.POINTER(sz)]
    L.blosc_cbuffer_metainfo.argtypes = [vp, C.POINTER(sz), C.POINTER(i)]
    L.blosc_cbuffer_metainfo.restype = None
    L.blosc_cbuffer_versions.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.blosc_cbuffer_versions.restype = None
    L.blosc_set_blocksize.argtypes = [sz]
    L.blosc_set_blocksize.restype = None
    L.blosc_set_splitmode.argtypes = [i]
    L.blosc_set_splitmode.restype = None
    L.blosc_gpu_compress_batch.argtypes = [i, i, sz, C.c_char_p, sz, i, C.POINTER(vp), C.POINTER(sz),
                                           C.POINTER(vp), C.POINTER(sz), C.POINTER(i), vp]
    L.blosc_gpu_decompress_batch.argtypes = [i, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz),
                                             C.POINTER(i), vp]
    L.blosc_gpu_compress_batch_host.argtypes = [i, i, sz, C.c_char_p, sz, i, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(i)]
    L.blosc_gpu_decompress_batch_host.argtypes = [i, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(i)]
    if hasattr(L, "blosc_gpu_partition"):        # (A/B scripts also load builds of earlier rounds through this loader)
        L.blosc_gpu_partition.argtypes = [sz, i, i, C.POINTER(sz), C.POINTER(sz)]
        L.blosc_gpu_compress_batch_multi.argtypes = [i, C.POINTER(i), i, i, sz, C.c_char_p, sz, i, C.POINTER(vp), C.POINTER(sz),
                                                     C.POINTER(vp), C.POINTER(sz), C.POINTER(i)]
        L.blosc_gpu_decompress_batch_multi.argtypes = [i, C.POINTER(i), i, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(i)]
    L.blosc_gpu_getitem.argtypes = [vp, i, i, vp, vp]
    L.blosc_gpu_profile.argtypes = [i]
    L.blosc_gpu_profile.restype = None
    L.blosc_gpu_profile_reset.restype = None
    L.blosc_gpu_profile_get.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(i)]
    for name in ("blosc_internal_shuffle", "blosc_internal_unshuffle"):
        getattr(L, name).argtypes = [sz, sz, vp, vp]
        getattr(L, name).restype = None
    for name in ("blosc_internal_bitshuffle", "blosc_internal_bitunshuffle"):
        getattr(L, name).argtypes = [sz, sz, vp, vp, vp]
    if hasattr(L, "blosc_amd_policy_blocksize"):
        L.blosc_amd_policy_blocksize.argtypes = [i, i, i, i, i, i]
        L.blosc_amd_policy_split.argtypes = [i, i, i, i]
    _lib = L
    return L


# ---- numpy conveniences (host buffers through the stock entry points) --------------------------
def compress(arr, typesize, clevel=5, shuffle=1, cname=b"lz4", blocksize=0, destsize=None):
    """blosc_compress_ctx on a numpy array; returns (return_code, bytes-or-None)."""
    import numpy as np
    L = load()
    a = np.ascontiguousarray(arr).view(np.uint8).ravel()
    cap = a.size + 16 if destsize is None else destsize
    out = np.empty(max(cap, 1), np.uint8)
    r = L.blosc_compress_ctx(clevel, shuffle, typesize, a.size, a.ctypes.data, out.ctypes.data, cap, cname, blocksize, 1)
    return r, (out[:r].copy() if r > 0 else None)


def decompress(chunk, nbytes):
    """blosc_decompress_ctx of a numpy uint8 chunk into a fresh array of nbytes."""
    import numpy as np
    L = load()
    c = np.ascontiguousarray(chunk)
    out = np.zeros(max(nbytes, 1), np.uint8)
    r = L.blosc_decompress_ctx(c.ctypes.data, out.ctypes.data, nbytes, 1)
    return r, out[:max(r, 0)]


class DeviceBatch:
    """Helper for the device-resident batched API: keeps the ctypes pointer/size arrays alive."""

    def __init__(self, src_ptrs, src_sizes, dst_ptrs, dst_sizes):
        n = len(src_ptrs)
        self.n = n
        self.src = (C.c_void_p * n)(*src_ptrs)
        self.dst = (C.c_void_p * n)(*dst_ptrs)
        self.ssz = (C.c_size_t * n)(*src_sizes)
        self.dsz = (C.c_size_t * n)(*dst_sizes)
        self.res = (C.c_int * n)()

    def compress(self, typesize, clevel=5, shuffle=1, cname=b"lz4", blocksize=0, stream=None):
        return load().blosc_gpu_compress_batch(clevel, shuffle, typesize, cname, blocksize, self.n, self.src,
                                               self.ssz, self.dst, self.dsz, self.res, stream)

    def decompress(self, stream=None, with_srcsize=True):
        return load().blosc_gpu_decompress_batch(self.n, self.src, self.ssz if with_srcsize else None, self.dst,
                                                 self.dsz, self.res, stream)

    def results(self):
        return list(self.res)


def profile_get(name):
    ms, cnt = C.c_double(0), C.c_int(0)
    load().blosc_gpu_profile_get(name.encode(), C.byref(ms), C.byref(cnt))
    return ms.value, cnt.value
