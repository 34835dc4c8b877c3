#!/usr/bin/env python3
"""Timeline of the loss phase of the last complete pair step (from the last forward kernel to the first backward kernel) out of a
rocprofv3 --kernel-trace CSV: start / end (us, relative), stream, kernel.  usage: loss_phase_timeline.py <dir>"""
import csv, glob, os, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(kt[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
begins = [i for i, e in enumerate(ev) if "step_begin_kernel" in e[2]]
durs = []
for a, b in zip(begins[:-1], begins[1:]):  # every complete step: the duration of its loss phase
    st = ev[a:b]
    try:
        j0 = max(i for i, e in enumerate(st) if "desc_normalize_kernel" in e[2])
        j1 = min(i for i, e in enumerate(st) if i > j0 and ("bn_bwd" in e[2] or "colsum" in e[2]))
        durs.append((st[j1][0] - st[j0][0]) / 1e3)
    except ValueError:
        pass
print("loss phase of every complete step (us):", " ".join("%.0f" % d for d in durs))
step = ev[begins[-2]:begins[-1]]
i0 = max(i for i, e in enumerate(step) if "desc_normalize_kernel" in e[2])
i1 = min(i for i, e in enumerate(step) if i > i0 and ("bn_bwd" in e[2] or "colsum" in e[2]))
t0 = step[i0][0]
print("loss phase: %.1f us from the start of the last forward kernel to the start of the first backward kernel" % ((step[i1][0] - t0) / 1e3))
for s, e, n, q in step[i0:i1 + 1]:
    print("%8.1f %8.1f  %6.1f us  stream %-4s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n.split("(")[0][:70]))
