"""Compile-time ablation of conv_wino_p2_kernel (P2_ABL bits: 1 no epilogue, 2 no halo staging / transform, 4 no barriers,
8 no MFMA, 16 no weight loads, 32 no LDS fragment reads).  build (no GPU needed): python tools/ablate_p2.py build ;
on the GPU box: python tools/ablate_p2.py run"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "semantic-superpoint_amd", "csrc")
VARIANTS = [int(v) for v in os.environ.get("ABL_VARIANTS", "0,1,2,3,4,16,32,48,51,55,8").split(",")]
if sys.argv[1] == "build":
    procs = []
    for v in VARIANTS:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
               "-DP2_ABL=%d" % v, "ssp.hip", "-o", "p2abl_%d.so" % v]
        procs.append(subprocess.Popen(cmd, cwd=CSRC))
        if len(procs) == 4:
            [p.wait() for p in procs]; procs = []
    [p.wait() for p in procs]
elif sys.argv[1] == "run":
    for v in VARIANTS:
        subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(v)])
else:
    sys.path.insert(0, ROOT)
    import torch
    from semantic_superpoint_amd import lib as L
    v = int(sys.argv[2])
    lib = L.load_library(os.path.join(CSRC, "p2abl_%d.so" % v))
    L.set_conv_algo(6)
    dev = torch.device("cuda:0")
    N, H, W, C = 32, 240, 320, 64
    x = torch.randn(N, H, W, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    b = torch.zeros(C, device=dev); sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
    res = []
    for grid in (256, 512):
        lib.ssp_debug_conv_knobs(0, grid)
        for mode in (1,):
            for _ in range(4): L.op_conv(x, w, b, 3, mode, sc, sh, None)
            torch.cuda.synchronize()
            t = []
            for _ in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); L.op_conv(x, w, b, 3, mode, sc, sh, None); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
            res.append("grid %d: %.3f ms" % (grid, sorted(t)[4]))
    print("P2_ABL=%3d  " % v + "  ".join(res), flush=True)
