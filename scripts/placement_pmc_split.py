"""Post-processing of scripts/placement_pmc.sh: join the counters of every k_decode_streams launch with its duration, split the launches at the middle
of the duration range, print every counter's mean in the two halves."""
import collections, csv, glob, os, sys
d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not cc: sys.exit("no counter file under " + d)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        if "k_decode_streams" in r.get("Kernel_Name", ""): dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc[0])):
    if "k_decode_streams" not in r.get("Kernel_Name", ""): continue
    did = r["Dispatch_Id"]
    if did not in dur and r.get("End_Timestamp"): dur[did] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    vals[r["Counter_Name"]][did] = vals[r["Counter_Name"]].get(did, 0.0) + float(r["Counter_Value"])
ids = sorted((k for k in dur if all(k in v for v in vals.values())), key=int)
if not ids: sys.exit(f"no launch with a duration ({len(dur)} durations, {sum(len(v) for v in vals.values())} counter rows)")
ds = [dur[k] for k in ids]; lo, hi = min(ds), max(ds); mid = (lo + hi) / 2
fast = [k for k in ids if dur[k] < mid]; slow = [k for k in ids if dur[k] >= mid]
mean = lambda ks, f: sum(f(k) for k in ks) / max(len(ks), 1)
print(f"   {len(ids)} launches, {lo:.3f} .. {hi:.3f} ms; fast half {len(fast)} launches avg {mean(fast, dur.get):.3f} ms, slow half {len(slow)} avg {mean(slow, dur.get):.3f} ms")
print("   durations in launch order: " + " ".join(f"{x:.2f}" for x in ds))
for c, v in vals.items():
    a, b = mean(fast, v.get), mean(slow, v.get)
    print(f"   {c:42s} fast {a:16.0f}   slow {b:16.0f}   slow/fast {b / a if a else float('nan'):.3f}")
