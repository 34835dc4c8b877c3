"""Perf-debug: time conv_mfma_kernel on the dominant layer shape with parts of the kernel disabled."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_superpoint_amd import lib as L
dev = torch.device("cuda:0")
N, H, W, C = 32, 240, 320, 64
x = torch.randn(N, H, W, C, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
b = torch.zeros(C, device=dev)
sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
flops = 2.0 * N * H * W * C * C * 9
GRID = [0]
if len(sys.argv) > 1:
    L.set_conv_algo(int(sys.argv[1]))  # 0 direct, 1 Winograd
def run(tag, mode, abl, stats=False):
    L.load_library().ssp_debug_conv_knobs(int(abl), GRID[0])
    st = torch.zeros(L.NREP, 2 * C, dtype=torch.float64, device=dev) if stats else None
    for _ in range(2):
        L.op_conv(x, w, b, 3, mode, sc, sh, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = []
    for _ in range(5):
        e0.record(); L.op_conv(x, w, b, 3, mode, sc, sh, st); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    ms = sorted(t)[len(t) // 2]
    print("%-40s %8.3f ms  %7.1f TF (includes pack + alloc overhead)" % (tag, ms, flops / ms / 1e9))
run("full (mode1, stats)", 1, 0, True)
run("full (mode0)", 0, 0)
run("no stores", 0, 4)
run("no global loads", 0, 1)
run("no global loads, no stores", 0, 5)
run("no loads/LDS writes/stores (MFMA+ds_read)", 0, 7)
run("no MFMA (loads+LDS+stores)", 0, 8)
run("no MFMA, no stores", 0, 12)
run("no global stores only (16)", 0, 16)
run("no LDS epilogue passes (32)", 0, 32)
run("no LDS epilogue, no global stores (48)", 0, 48)

for g in ((256, 512, 768, 1024) if len(sys.argv) < 2 or sys.argv[1] == "0" else ()):
    GRID[0] = g
    run("full mode0, grid %d" % g, 0, 0)
    run("MFMA only, grid %d" % g, 0, 7)
