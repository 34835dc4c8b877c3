#!/usr/bin/env python3
"""Timeline of one data-parallel pair step from rocprofv3 CSV traces (kernel_trace + memory_copy_trace) of ONE rank:
shows the early gradient bucket leaving the device (gloo on a 1-GPU box: a device-to-host copy; RCCL on a real node: its own
kernels) while the phase-2 kernels (backward of the two 240x320 layers) are still being issued / running.
usage: overlap_timeline.py <dir with *_kernel_trace.csv / *_memory_copy_trace.csv>"""
import csv, glob, os, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
mc = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
ev = []
for r in csv.DictReader(open(kt[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"]))
if mc:
    rows = list(csv.DictReader(open(mc[0])))
    if rows:
        print("memory-copy trace columns:", ", ".join(rows[0].keys()))
    for r in rows:
        size = next((r[k] for k in ("Bytes", "Size", "Size_Bytes") if k in r), "?")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", "%s %s bytes" % (r.get("Direction", "copy"), size)))
ev.sort()
begins = [i for i, e in enumerate(ev) if e[2] == "K" and "step_begin_kernel" in e[3]]
lo, hi = begins[-2], begins[-1]  # the last complete step
t0 = ev[lo][0]
step = ev[lo:hi]
copies = [e for e in step if e[2] == "C"]
kern = [e for e in step if e[2] == "K"]
adam = [e for e in kern if "adam" in e[3]]
red = [e for e in kern if "wgrad_wino_reduce_multi_kernel" in e[3]]  # phase 1 ends with its slab reduction, phase 2 with another
print("step wall %.3f ms, %d kernels, %d copies (columns: no byte count in this rocprofv3 version)" % ((ev[hi][0] - t0) / 1e6, len(kern), len(copies)))
for e in copies:
    print("copy      start %8.3f ms  dur %7.1f us  %s" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e3, e[3]))
if red:
    t1 = red[0][1]
    print("phase 1 ends (early bucket final, all-reduce of grads[early:] issued) at %8.3f ms" % ((t1 - t0) / 1e6))
    p2 = [e for e in kern if e[0] >= t1 and (not adam or e[0] < adam[0][0])]
    if p2:
        print("phase 2 (backward of the 240x320 layers): %d kernels, %.3f ms of GPU time, from %.3f to %.3f ms"
              % (len(p2), sum(e[1] - e[0] for e in p2) / 1e6, (p2[0][0] - t0) / 1e6, (p2[-1][1] - t0) / 1e6))
        for e in p2[:4] + p2[-2:]:
            print("   %8.3f ms  %7.1f us  %s" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e3, e[3][:90]))
    h2d = [e for e in copies if "HOST_TO_DEVICE" in e[3] and e[0] >= t1]
    if h2d and p2:
        print("the reduced early bucket comes back (host-to-device copy of the gloo result) at %.3f ms: %.3f ms after phase 1 ended, "
              "%.3f ms of which the GPU spent on phase-2 kernels" % ((h2d[0][0] - t0) / 1e6, (h2d[0][0] - t1) / 1e6,
                                                                  sum(min(e[1], h2d[0][0]) - e[0] for e in p2 if e[0] < h2d[0][0]) / 1e6))
if adam:
    print("adam      start %8.3f ms" % ((adam[0][0] - t0) / 1e6))
