"""Does the speed of the Winograd conv depend on the DATA (power / clock management)?  Same launch, same instruction
stream, different operand values.  64->64 @240x320, N = 32 (one view)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_superpoint_amd import lib as L
dev = torch.device("cuda:0")
N, H, W, C = 32, 240, 320, 64
b = torch.zeros(C, device=dev); sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
def t(x, w, tag):
    for _ in range(4): L.op_conv(x, w, b, 3, 0, sc, sh, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(15):
        e0.record(); L.op_conv(x, w, b, 3, 0, sc, sh, None); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[7]
    print("%-44s %.3f ms  %6.1f algorithmic TF" % (tag, ms, 2.0 * N * H * W * C * C * 9 / ms / 1e9), flush=True)
xr = torch.randn(N, H, W, C, device=dev); wr = torch.randn(C, C, 3, 3, device=dev) * 0.05
t(xr, wr, "x = randn, w = randn")
t(torch.relu(xr), wr, "x = relu(randn) (half zeros), w = randn")
t(torch.zeros_like(xr), wr, "x = 0, w = randn")
t(xr, torch.zeros_like(wr), "x = randn, w = 0")
t(torch.zeros_like(xr), torch.zeros_like(wr), "x = 0, w = 0")
t(torch.ones_like(xr), wr, "x = 1 (constant), w = randn")
t(xr, wr, "x = randn, w = randn (again)")
