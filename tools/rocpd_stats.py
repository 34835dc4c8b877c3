#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) as per-kernel statistics (like --stats CSV)."""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = ("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.%s "
         "order by 3 desc" % (name_col, name_col))
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows)
    print("%-90s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
    for name, n, t, mn, mx in rows[:top]:
        print("%-90s %7d %12.3f %10.1f %10.1f %10.1f %6.2f" % (name[:90], n, t / 1e6, t / n / 1e3, mn / 1e3, mx / 1e3,
                                                                100.0 * t / tot))
    print("TOTAL kernel time: %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
