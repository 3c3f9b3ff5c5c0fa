import sys, os, numpy as np, ctypes as C, importlib.util
ROOT='/root/repo'; sys.path.insert(0, ROOT+'/tests')
from helpers import *
spec = importlib.util.spec_from_file_location("c_blosc_amd", ROOT+"/c-blosc_amd/__init__.py"); pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
for n in [128, 1000]:
    data = DATASETS['bench19'](n)
    r, chunk = pkg.compress(data, 1, 1, 0, b'lz4')
    print(n, r, np.array_equal(chunk[16:], data), list(chunk[16:40]))
