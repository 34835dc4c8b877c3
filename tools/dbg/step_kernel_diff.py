#!/usr/bin/env python3
"""Per-kernel time per step of two rocprofv3 --kernel-trace CSV directories, side by side, and the wall span of a step (first
launch of a step to the first launch of the next).  usage: step_kernel_diff.py <dirA> <dirB>"""
import csv, glob, os, sys, collections, re


def load(d):
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(kt[0])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    begins = [i for i, e in enumerate(ev) if "step_begin_kernel" in e[2]]
    tot = collections.Counter()
    cnt = collections.Counter()
    n = len(begins) - 1
    spans = []
    for a, b in zip(begins[:-1], begins[1:]):
        spans.append((ev[b][0] - ev[a][0]) / 1e3)
        for s, e, k in ev[a:b]:
            k = re.sub(r"\(.*", "", k)
            k = k.replace("void ", "").replace("sspk::", "")
            tot[k] += (e - s) / 1e3 / n
            cnt[k] += 1.0 / n
    return tot, cnt, spans


ta, ca, sa = load(sys.argv[1])
tb, cb, sb = load(sys.argv[2])
print("step span us  A:", " ".join("%.0f" % x for x in sa))
print("step span us  B:", " ".join("%.0f" % x for x in sb))
keys = sorted(set(ta) | set(tb), key=lambda k: -(max(ta[k], tb[k])))
print("%-70s %10s %10s %9s   launches A / B" % ("kernel", "A us/step", "B us/step", "B - A"))
for k in keys:
    if max(ta[k], tb[k]) < 3 and abs(tb[k] - ta[k]) < 3:
        continue
    print("%-70s %10.1f %10.1f %9.1f   %.0f / %.0f" % (k[:70], ta[k], tb[k], tb[k] - ta[k], ca[k], cb[k]))
print("%-70s %10.1f %10.1f %9.1f" % ("sum of kernel times", sum(ta.values()), sum(tb.values()), sum(tb.values()) - sum(ta.values())))
