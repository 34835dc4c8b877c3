#!/usr/bin/env python3
"""Times the 3x3 convolution operator (ssp_op_conv) per algorithm on the shapes of the network.
usage: python tools/conv_probe.py [algos=1,6] [reps=9]   (run on the GPU box); SSP_PROBE_SHAPES=big restricts to 64->64@240x320"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from semantic_superpoint_amd import lib as L

algos = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,6").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
dev = torch.device("cuda:0")
SHAPES = [(32, 240, 320, 64, 64), (32, 120, 160, 64, 64), (32, 60, 80, 64, 128), (32, 60, 80, 128, 128), (32, 30, 40, 128, 128),
          (32, 30, 40, 128, 256)]
if os.environ.get("SSP_PROBE_SHAPES") == "big":
    SHAPES = SHAPES[:1]
for (N, H, W, ci, co) in SHAPES:
    x = torch.randn(N, H, W, ci, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    b = torch.zeros(co, device=dev); sc = torch.ones(ci, device=dev); sh = torch.zeros(ci, device=dev)
    gf = 2.0 * N * H * W * ci * co * 9 / 1e9
    for mode in (0, 1):
        for with_stats in (True, False):
            row = []
            for algo in algos:
                L.set_conv_algo(algo)
                st = torch.zeros(L.NREP, 2 * co, dtype=torch.float64, device=dev) if with_stats else None
                for _ in range(3):
                    L.op_conv(x, w, b, 3, mode, sc, sh, st)
                torch.cuda.synchronize()
                ts = []
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); L.op_conv(x, w, b, 3, mode, sc, sh, st); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                t = sorted(ts)[len(ts) // 2]
                row.append("algo %d: %.3f ms (%.0f TF alg)" % (algo, t, gf / t))
            print("%dx%dx%d %d->%d mode %d stats %d | " % (N, H, W, ci, co, mode, with_stats) + " | ".join(row), flush=True)
L.set_conv_algo(1)
if os.environ.get("SSP_PROBE_GRID"):
    lib = L.load_library()
    print("occupancy (blocks/CU): p2 %d, pipe %d, wgrad_wino %d" % (lib.ssp_debug_occupancy(0), lib.ssp_debug_occupancy(1), lib.ssp_debug_occupancy(2)))
    N, H, W, ci, co = 32, 240, 320, 64, 64
    x = torch.randn(N, H, W, ci, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    b = torch.zeros(co, device=dev); sc = torch.ones(ci, device=dev); sh = torch.zeros(ci, device=dev)
    L.set_conv_algo(6)
    for grid in (128, 256, 512, 768, 1024):
        lib.ssp_debug_conv_knobs(0, grid)
        for _ in range(3):
            L.op_conv(x, w, b, 3, 1, sc, sh, None)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); L.op_conv(x, w, b, 3, 1, sc, sh, None); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("algo 6 grid %4d: %.3f ms" % (grid, sorted(ts)[3]), flush=True)
    lib.ssp_debug_conv_knobs(0, 0)
    L.set_conv_algo(1)
