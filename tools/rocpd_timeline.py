#!/usr/bin/env python3
"""Dispatch timeline of the LAST step in a rocprofv3 rocpd database: every kernel launch in order with its duration and the
idle gap before it.  usage: rocpd_timeline.py results.db [substring filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
scols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in scols else "display_name"
rows = list(c.execute("select s.%s, d.start, d.end, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start" % name_col))
# the last step starts at the last step_begin_kernel
starts = [i for i, r in enumerate(rows) if "step_begin_kernel" in r[0]]
lo = starts[-1] if starts else 0
flt = sys.argv[2] if len(sys.argv) > 2 else ""
prev_end = rows[lo][1]
tot = 0
for name, st, en, gx, wx in rows[lo:]:
    if "adam" in name and tot > 0: last = True
    if flt in name:
        print("%9.1f us  gap %6.1f  blocks %6d  %s" % ((en - st) / 1e3, (st - prev_end) / 1e3, gx // max(wx, 1), name[:100]))
    tot += en - st
    prev_end = en
print("step: %d launches, kernel time %.3f ms, wall %.3f ms" % (len(rows) - lo, tot / 1e6, (rows[-1][2] - rows[lo][1]) / 1e6))
