"""Compile-time ablation of conv_wino_pipe_kernel (PIPE_ABL bits: 1 no epilogue, 2 no staging, 4 no barriers, 8 no MFMA).
Build the variants here (no GPU needed):  python tools/ablate_pipe.py build
Time them on the GPU box:                 python tools/ablate_pipe.py run"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "semantic-superpoint_amd", "csrc")
VARIANTS = [int(v) for v in os.environ.get("ABL_VARIANTS", "0,1,2,3,6,7,8,15").split(",")]
if sys.argv[1] == "build":
    procs = []
    for v in VARIANTS:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
               "-DPIPE_ABL=%d" % v, "ssp.hip", "-o", "abl_%d.so" % v]
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
    L.load_library(os.path.join(CSRC, "abl_%d.so" % v))
    dev = torch.device("cuda:0")
    N, H, W, C = 32, 240, 320, 64
    x = torch.randn(N, H, W, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    b = torch.zeros(C, device=dev); sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
    st = torch.zeros(L.NREP, 2 * C, dtype=torch.float64, device=dev)
    res = []
    for mode in (0, 1):
        for _ in range(4): L.op_conv(x, w, b, 3, mode, sc, sh, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); t = []
        for _ in range(9):
            e0.record(); L.op_conv(x, w, b, 3, mode, sc, sh, st); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
        res.append(sorted(t)[4])
    print("PIPE_ABL=%2d  mode0 %.3f ms  mode1 %.3f ms" % (v, res[0], res[1]), flush=True)
