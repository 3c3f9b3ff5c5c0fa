import sys, os, numpy as np, ctypes as C, importlib.util
ROOT='/root/repo'; sys.path.insert(0, ROOT+'/tests')
from helpers import *
spec = importlib.util.spec_from_file_location("c_blosc_amd", ROOT+"/c-blosc_amd/__init__.py"); pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
O = C.CDLL(ROOT+'/oracle/liboracle.so'); O.orc_compress.argtypes=[C.c_int,C.c_int,C.c_size_t,C.c_size_t,C.c_void_p,C.c_void_p,C.c_size_t,C.c_int,C.c_size_t,C.c_int]
def parse(s):
    ip=0; seqs=[]; op=0
    while ip < len(s):
        t0=ip; t=s[ip]; ip+=1; ll=t>>4
        if ll==15:
            while True:
                b=s[ip]; ip+=1; ll+=b
                if b!=255: break
        ip+=ll
        if ip>=len(s): seqs.append((t0,op,ll,0,0)); break
        off=s[ip]|(s[ip+1]<<8); ip+=2; ml=t&15
        if ml==15:
            while True:
                b=s[ip]; ip+=1; ml+=b
                if b!=255: break
        seqs.append((t0,op,ll,off,ml+4)); op+=ll+ml+4
    return seqs
import itertools
for dn, n, who in itertools.product(['randwalk','zeros','smallints','bench19'], [4096, 32768], ['oracle','gpu']):
    data = DATASETS[dn](n)
    if who=='oracle': r, chunk = orc_compress(O, data, 1, 1, 0, 'lz4')
    else: r, chunk = pkg.compress(data, 1, 1, 0, b'lz4')
    r2, out = pkg.decompress(chunk, n)
    bad = np.nonzero(out != data)[0]
    print(dn, n, who, 'cbytes', r, 'decode', r2, 'mismatches', bad.size, bad[:10])
    st = chunk[24:].tolist()
    sq = parse(st)
    print(' nseq', len(sq))
    if bad.size:
        b0=bad[0]
        for q in sq:
            if q[1] <= b0 < q[1]+q[2]+q[4]+64: print('  seq tokpos %d op %d ll %d off %d ml %d'%q)
        print(' got ', out[b0-4:b0+12].tolist()); print(' want', data[b0-4:b0+12].tolist())
