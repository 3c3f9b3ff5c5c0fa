import sys, os, numpy as np, ctypes as C, importlib.util, time
ROOT='/root/repo'; sys.path.insert(0, ROOT+'/tests')
from helpers import *
spec = importlib.util.spec_from_file_location("c_blosc_amd", ROOT+"/c-blosc_amd/__init__.py"); pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
O = C.CDLL(ROOT+'/oracle/liboracle.so'); O.orc_compress.argtypes=[C.c_int,C.c_int,C.c_size_t,C.c_size_t,C.c_void_p,C.c_void_p,C.c_size_t,C.c_int,C.c_size_t,C.c_int]; O.orc_decompress.argtypes=[C.c_void_p,C.c_void_p,C.c_size_t]
for n in [1000, 300001, (1<<22)+24, 1<<25]:
    for codec in ['lz4','blosclz']:
        data = DATASETS['bench19'](n)
        t0=time.time(); ro, oc = orc_compress(O, data, 8, 5, 1, codec)
        r, out = pkg.decompress(oc, n); t1=time.time()
        rc, ch = pkg.compress(data, 8, 5, 1, codec.encode()); t2=time.time()
        r2, o2 = orc_decompress(O, ch, n) if rc>0 else (-1,None)
        print(n, codec, 'dec', r==n and np.array_equal(out,data), 'enc', rc, r2==n and np.array_equal(o2,data), f'{t1-t0:.2f}s {t2-t1:.2f}s', flush=True)
