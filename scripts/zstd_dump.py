"""GPU-compress one buffer with zstd and save the chunk (for offline inspection of the frames)."""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATASETS
spec = importlib.util.spec_from_file_location("c_blosc_amd", os.path.join(ROOT, "c-blosc_amd", "__init__.py")); pkg = importlib.util.module_from_spec(spec); spec.loader.exec_module(pkg)
pkg.load()
for dname in ["bench19", "linspace"]:
    data = DATASETS[dname](4 << 20)
    r, ch = pkg.compress(data, 8, 3, 1, b"zstd")
    ch.tofile(os.path.join(ROOT, "gpurun_out", f"zstd_chunk_{dname}.bin"))
    print(dname, r)
