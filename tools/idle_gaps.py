#!/usr/bin/env python3
"""Idle time of the GPU inside one pair step, from a rocprofv3 --kernel-trace CSV: union of the busy intervals of all
streams over the last complete step (step_begin_kernel .. step_begin_kernel), and the gaps grouped by the kernel that
FOLLOWS each gap (the launch that arrived late).
usage: idle_gaps.py <dir with *_kernel_trace.csv> [top n]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt[0])))
begins = [i for i, e in enumerate(ev) if "step_begin_kernel" in e[2]]
lo, hi = begins[-2], begins[-1]
step = ev[lo:hi]
t0, t1 = step[0][0], ev[hi][0]
busy_end = step[0][0]
idle = 0
gaps = collections.defaultdict(lambda: [0, 0])
overlap = 0
for s, e, n in step:
    if s > busy_end:
        g = s - busy_end
        idle += g
        k = n.split("(")[0][:70]
        gaps[k][0] += g
        gaps[k][1] += 1
    else:
        overlap += min(e, busy_end) - s
    busy_end = max(busy_end, e)
idle += max(0, t1 - busy_end)
print("step wall %.3f ms, %d kernels, kernel time %.3f ms, concurrent (second stream) %.3f ms, idle %.3f ms (%.1f %%)"
      % ((t1 - t0) / 1e6, len(step), sum(e - s for s, e, _ in step) / 1e6, overlap / 1e6, idle / 1e6, 100.0 * idle / (t1 - t0)))
print("gap before kernel                                                        total us   count   avg us")
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%-72s %8.1f %7d %8.2f" % (k, g / 1e3, c, g / 1e3 / c))
