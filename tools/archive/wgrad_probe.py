#!/usr/bin/env python3
"""Times the 3x3 weight-gradient operator (ssp_op_conv_wgrad, incl. its slab reduction) in the sustained regime.
usage: python tools/wgrad_probe.py [tag] [shapes=big|all]   (GPU box; SSP_HIP_LIB / SSP_WGRAD_F4 select the build / kernel)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from semantic_superpoint_amd import lib as L
tag = sys.argv[1] if len(sys.argv) > 1 else ""
shapes = [(32, 240, 320, 64, 64)]
if len(sys.argv) > 2 and sys.argv[2] == "all":
    shapes += [(32, 120, 160, 64, 64), (32, 60, 80, 64, 128), (32, 60, 80, 128, 128), (32, 30, 40, 128, 128), (32, 30, 40, 128, 256)]
dev = torch.device("cuda:0")
out = []
for (N, H, W, ci, co) in shapes:
    x = torch.randn(N, H, W, ci, device=dev); dy = torch.randn(N, H, W, co, device=dev)
    sc = torch.ones(ci, device=dev); sh = torch.zeros(ci, device=dev)
    for _ in range(20): L.op_conv_wgrad(x, dy, 3, 1, sc, sh)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60): L.op_conv_wgrad(x, dy, 3, 1, sc, sh)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 60
    out.append("%dx%d %d->%d %.3f ms (%.0f TF alg)" % (H, W, ci, co, ms, 2.0 * N * H * W * ci * co * 9 / ms / 1e9))
print("%-14s %s" % (tag, " | ".join(out)), flush=True)
