#!/usr/bin/env python3
"""Per-kernel HBM traffic and bandwidth from the FETCH_SIZE / WRITE_SIZE PMC summaries (tools/pmc_summary.py output):
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE doubled on gfx950: MI355X_MICROARCH.md), GB/s = bytes / kernel time.
usage: hbm_table.py <pmc_fetch_summary.txt> <pmc_write_summary.txt> [min_us]"""
import sys
def load(f):
    d = {}
    for l in open(f).read().split('\n')[1:]:
        if not l.strip(): continue
        name = l[:60].strip(); p = l[60:].split()
        d[name] = (int(p[0]), float(p[1]), float(p[2]))
    return d
fe, wr = load(sys.argv[1]), load(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
print("%-60s %5s %9s %10s %10s %8s %6s" % ("kernel", "calls", "avg us", "read MB", "write MB", "TB/s", "of 8"))
for k in sorted(fe, key=lambda k: -fe[k][1]):
    if k not in wr: continue
    calls, ms, f = fe[k]; w = wr[k][2]; ms = 0.5 * (ms + wr[k][1])
    if ms / calls * 1e3 < min_us: continue
    rd, wb = 2 * f * 1024 / calls, w * 1024 / calls
    tbs = (rd + wb) * calls / (ms * 1e-3) / 1e12
    print("%-60s %5d %9.1f %10.1f %10.1f %8.2f %6.2f" % (k, calls, ms / calls * 1e3, rd / 1e6, wb / 1e6, tbs, tbs / 8.0))
