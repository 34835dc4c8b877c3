"""Perf-debug: time the (Winograd) weight-gradient kernel on the dominant layer shape with parts disabled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_superpoint_amd import lib as L
dev = torch.device("cuda:0")
N, H, W, C = 32, 240, 320, 64
if len(sys.argv) > 1:
    L.set_conv_algo(int(sys.argv[1]))
x = torch.randn(N, H, W, C, device=dev); dy = torch.randn(N, H, W, C, device=dev)
sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
flops = 2.0 * N * H * W * C * C * 9
def run(tag, mode, abl):
    # sustained regime: 100 back-to-back launches after 30 warm-up launches (short bursts depend on the operand data / clocks)
    L.load_library().ssp_debug_conv_knobs(int(abl), 0)
    for _ in range(30): L.op_conv_wgrad(x, dy, 3, mode, sc, sh)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): L.op_conv_wgrad(x, dy, 3, mode, sc, sh)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print("%-40s %8.3f ms  %7.1f TF (includes reduce + allocs)" % (tag, ms, flops / ms / 1e9), flush=True)
run("full mode1", 1, 0); run("full mode0", 0, 0); run("no global loads", 0, 1); run("no loads, no LDS writes", 0, 3)
run("no MFMA loop", 0, 8); run("no MFMA, no loads", 0, 9); run("full mode1 again", 1, 0)
L.load_library().ssp_debug_conv_knobs(0, 0)
