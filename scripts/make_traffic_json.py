#!/usr/bin/env python3
"""profiles/<tag>_pmc_{FETCH,WRITE}_SIZE.csv -> profiles/<tag>_traffic.json (HBM bytes per launch and kernel).

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are
in KiB; FETCH_SIZE reports half of the bytes of a wide streaming read, so it is doubled; WRITE_SIZE is
taken as it is.  Both factors are checked against the torch kernels of known size in the same passes
(1 GiB fill: WRITE_SIZE = 1048576 KiB; 1 GiB == 1 GiB compare: FETCH_SIZE = 1048623 KiB for 2 GiB read).
"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
def rd(counter):
    out = {}
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{counter}.csv")) as fh:
        for row in csv.DictReader(fh):
            out[row["kernel"]] = float(row["mean_value_KB"])
    return out
f, w = rd("FETCH_SIZE"), rd("WRITE_SIZE")
res = {"workload": "bench.py defaults: 128 x 64 MiB bench19 chunks, byte-shuffle + lz4 clevel 5 typesize 8",
       "corrections": {"FETCH_SIZE": "KiB x 1024 x 2", "WRITE_SIZE": "KiB x 1024"}, "kernels": {}}
for k in sorted(set(f) | set(w)):
    if not k.startswith("bamd::"): continue
    fb = f.get(k, 0.0) * 1024 * 2; wb = w.get(k, 0.0) * 1024
    res["kernels"][k.replace("bamd::", "")] = {"hbm_read_bytes": fb, "hbm_write_bytes": wb, "hbm_bytes": fb + wb}
json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
