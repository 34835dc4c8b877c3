#!/bin/bash
# Shader clock and socket power while the Winograd conv runs on random vs zero operands (same instruction stream):
# evidence for DESIGN.md section 8 (the fp32 step is power-managed: the clock follows the operand data).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/conv_power
mkdir -p $O
( for i in $(seq 1 400); do echo "t=$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Socket Graphics Package Power|Average Graphics Package Power' | tr '\n' ' ')"; sleep 0.05; done ) > $O/smi.log &
SMI=$!
python3 - <<PY > $O/phases.log 2>&1
import os, sys, time
sys.path.insert(0, "$R")
import torch
from semantic_superpoint_amd import lib as L
dev = torch.device("cuda:0")
N, H, W, C = 32, 240, 320, 64
b = torch.zeros(C, device=dev); sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
xr = torch.randn(N, H, W, C, device=dev); wr = torch.randn(C, C, 3, 3, device=dev) * 0.05
for tag, x, w in (("randn", xr, wr), ("zeros", torch.zeros_like(xr), wr), ("randn", xr, wr), ("zeros", torch.zeros_like(xr), wr)):
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(50): L.op_conv(x, w, b, 3, 0, sc, sh, None)
        n += 50; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    print("phase %s start %.3f end %.3f  %.3f ms per launch" % (tag, t0, time.time(), e0.elapsed_time(e1) / n), flush=True)
PY
kill $SMI 2>/dev/null
cat $O/phases.log | grep phase
python3 - <<PY
import re
ph = [l.split() for l in open("$O/phases.log") if l.startswith("phase")]
rows = []
for l in open("$O/smi.log"):
    m = re.match(r"t=([\d.]+) (.*)", l)
    if not m: continue
    t = float(m.group(1)); s = m.group(2)
    clk = re.search(r"sclk.*?\((\d+)Mhz\)", s); pw = re.search(r"Power \(W\): ([\d.]+)", s)
    rows.append((t, int(clk.group(1)) if clk else None, float(pw.group(1)) if pw else None))
for p in ph:
    t0, t1 = float(p[3]), float(p[5])
    sel = [r for r in rows if t0 + 1.0 < r[0] < t1]
    ck = [r[1] for r in sel if r[1]]; pw = [r[2] for r in sel if r[2]]
    print(p[1], "samples", len(sel), "sclk MHz mean %s" % (sum(ck) / len(ck) if ck else None), "power W mean %s" % (sum(pw) / len(pw) if pw else None))
PY
head -3 $O/smi.log
