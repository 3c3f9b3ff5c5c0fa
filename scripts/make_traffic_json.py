#!/usr/bin/env python3
"""profiles/<tag>_cfg<C>_pmc_{FETCH,WRITE}_SIZE.csv -> profiles/<tag>_traffic_cfg<C>.json (HBM bytes per launch and kernel).
usage: make_traffic_json.py [tag=r02] [config=2]

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are
in KiB; FETCH_SIZE reports half of the bytes of a wide streaming read, so it is doubled; WRITE_SIZE is
taken as it is.  Both factors are checked against the torch kernels of known size in the same passes
(the fill / copy kernels bench.py uses to set up its buffers).
"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
cfg = sys.argv[2] if len(sys.argv) > 2 else "2"
def rd(counter):
    out = {}
    with open(os.path.join(ROOT, "profiles", f"{tag}_cfg{cfg}_pmc_{counter}.csv")) as fh:
        for row in csv.DictReader(fh):
            out[row["kernel"]] = float(row["mean_value_KB"])
    return out
f, w = rd("FETCH_SIZE"), rd("WRITE_SIZE")
fp_file = os.path.join(ROOT, "profiles", f"{tag}_cfg{cfg}_src_fingerprint.txt")
res = {"workload": f"bench.py --config {cfg} (128 x 64 MiB chunks per GPU), one step",
       "src_fingerprint": open(fp_file).read().strip() if os.path.exists(fp_file) else None,      # bench.src_fingerprint() on the box that ran the passes
       "corrections": {"FETCH_SIZE": "KiB x 1024 x 2", "WRITE_SIZE": "KiB x 1024"}, "kernels": {}}
for k in sorted(set(f) | set(w)):
    if "bamd::" not in k: continue
    name = k.split("bamd::")[-1]
    # kernels that bench.py reports under the name of their pipeline stage
    if name.startswith("k_encode_streams_t<"):          # k_encode.hip: ENC_LZ 0, ENC_ZSTD 1, ENC_ZLIB 2, ENC_HC 3, ENC_ZSTD_T 4, ENC_ZSTD_HC 5, ENC_ZLIB_HC 6, ENC_ZSTD_TH 7, ENC_ZSTD_HCH 8, ENC_ZLIB_DYN 9, ENC_ZLIB_DYN_HC 10
        mode = name[len("k_encode_streams_t<"):].split(">")[0]
        name = {"true": "k_zstd_encode", "1": "k_zstd_encode", "4": "k_zstd_encode", "5": "k_zstd_encode", "7": "k_zstd_encode", "8": "k_zstd_encode",
                "2": "k_zlib_encode", "6": "k_zlib_encode", "9": "k_zlib_encode", "10": "k_zlib_encode", "3": "k_lz4hc_encode"}.get(mode, "k_encode_streams")
    elif name.startswith("k_bitfilter_fast<0>"): name = "k_bitshuffle"
    elif name.startswith("k_bitfilter_fast<1>"): name = "k_bitunshuffle"
    name = name.split("<")[0]
    fb = f.get(k, 0.0) * 1024 * 2; wb = w.get(k, 0.0) * 1024
    e = res["kernels"].setdefault(name, {"hbm_read_bytes": 0.0, "hbm_write_bytes": 0.0, "hbm_bytes": 0.0})
    e["hbm_read_bytes"] += fb; e["hbm_write_bytes"] += wb; e["hbm_bytes"] += fb + wb
# decode of REFERENCE-written chunks (the drop-in direction), counted apart: scripts/gpu_call.sh dectraffic runs the same two PMC passes over
# scripts/dec_sweep.py, whose k_decode_* launches all decode stock chunks (profiles/<tag>_dec_traffic.txt: "<COUNTER> <kernel>: launches n mean X M units")
stock_txt = os.path.join(ROOT, "profiles", f"{tag}_cfg{cfg}_dec_traffic_stock.txt")
if os.path.exists(stock_txt):
    ks = {}
    for ln in open(stock_txt):
        p = ln.split()
        if len(p) >= 6 and p[0] in ("FETCH_SIZE", "WRITE_SIZE") and p[4] == "mean":
            name = p[1].rstrip(":").split("bamd::")[-1].split("<")[0].split("(")[0]
            kib = float(p[5]) * 1e6
            e = ks.setdefault(name, {"hbm_read_bytes": 0.0, "hbm_write_bytes": 0.0, "hbm_bytes": 0.0})
            if p[0] == "FETCH_SIZE": e["hbm_read_bytes"] = kib * 1024 * 2
            else: e["hbm_write_bytes"] = kib * 1024
            e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    res["kernels_stock"] = ks
    res["kernels_stock_note"] = "launches of scripts/dec_sweep.py: reference-written chunks only (bench.py's own passes above decode chunks written here)"
json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_traffic_cfg{cfg}.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
