#!/usr/bin/env python3
"""Per-kernel PMC summary from rocprofv3 counter_collection.csv (+ kernel_trace.csv for durations)."""
import csv, sys, collections
cc, kt = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); cnt[k] += 1
        agg[k]["_ns"] += dur.get(r["Dispatch_Id"], (0, ""))[0]
names = sorted({c for k in agg for c in agg[k] if not c.startswith("_")})
rows = sorted(agg.items(), key=lambda kv: -kv[1]["_ns"])
print("%-60s %5s %9s " % ("kernel", "calls", "ms") + " ".join("%14s" % n[-14:] for n in names))
for k, v in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%-60s %5d %9.3f " % (k[:60], cnt[k], v["_ns"] / 1e6) + " ".join("%14.4g" % v[n] for n in names))
