"""Perf-debug / accuracy probe of the bf16-operand Winograd conv (ssp_set_conv_algo(3)) against fp32 torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from semantic_superpoint_amd import lib as L
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
for (N, H, W, cin, cout, mode) in ((2, 16, 64, 64, 64, 1), (2, 30, 40, 128, 256, 0), (1, 9, 11, 64, 70, 1)):
    x = torch.from_numpy(rs.randn(N, H, W, cin).astype(np.float32))
    w = torch.from_numpy((rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)).astype(np.float32))
    b = torch.from_numpy(rs.randn(cout).astype(np.float32) * 0.1)
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, cin).astype(np.float32)); sh = torch.from_numpy(rs.uniform(-0.5, 0.5, cin).astype(np.float32))
    xin = x.permute(0, 3, 1, 2)
    if mode: xin = F.relu(xin * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref = F.conv2d(xin, w, b, padding=1).permute(0, 2, 3, 1)
    for algo in (1, 3):
        L.set_conv_algo(algo)
        out = L.op_conv(x.to(dev), w.to(dev), b.to(dev), 3, mode, sc.to(dev), sh.to(dev), None).cpu()
        print("shape", (N, H, W, cin, cout, mode), "algo", algo, "rel max err %.2e  rel rms err %.2e" % (
            float((out - ref).abs().max() / ref.abs().max()), float((out - ref).norm() / ref.norm())))
N, H, W, C = 32, 240, 320, 64
x = torch.randn(N, H, W, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
for algo in (1, 3):
    L.set_conv_algo(algo)
    for _ in range(2): L.op_conv(x, w, None, 3, 0, None, None, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); t = []
    for _ in range(7):
        e0.record(); L.op_conv(x, w, None, 3, 0, None, None, None); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
    print("algo", algo, "64->64 @240x320 x32: %.3f ms" % sorted(t)[3])
L.set_conv_algo(1)
# weight gradient
x = torch.randn(N, H, W, C, device=dev); dy = torch.randn(N, H, W, C, device=dev)
ref = None
for algo in (1, 3):
    L.set_conv_algo(algo)
    for _ in range(2): dw = L.op_conv_wgrad(x, dy, 3, 0, None, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); t = []
    for _ in range(7):
        e0.record(); L.op_conv_wgrad(x, dy, 3, 0, None, None); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
    if ref is None: ref = dw
    print("wgrad algo", algo, "%.3f ms   rel rms vs fp32 Winograd %.2e" % (sorted(t)[3], float((dw - ref).norm() / ref.norm())))
L.set_conv_algo(1)
