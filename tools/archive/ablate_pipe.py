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
elif sys.argv[1] == "trace":
    # PIPE_ABL=1024 build: per-wave cycle stamps of block 8 at the stage boundaries (stage start, before / after the
    # mid-stage barrier, before the end-of-stage barrier)
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from semantic_superpoint_amd import lib as L
    L.load_library(os.path.join(CSRC, "abl_1024.so"))
    dev = torch.device("cuda:0")
    N, H, W, C = 32, 240, 320, 64
    x = torch.randn(N, H, W, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    b = torch.zeros(C, device=dev); sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
    for mode in (0, 1):
        st = torch.zeros(L.NREP, 2 * C, dtype=torch.float64, device=dev)
        for _ in range(3): L.op_conv(x, w, b, 3, mode, sc, sh, st)
        st.zero_(); L.op_conv(x, w, b, 3, mode, sc, sh, st); torch.cuda.synchronize()
        t = st.cpu().numpy().view(np.int64).reshape(-1)[:100 * 8 * 4].reshape(100, 8, 4)
        t0 = t[:, :, 0]
        print("mode %d: stage period (wave 0), stages 8..40:" % mode, np.diff(t0[8:41, 0]).tolist())
        for g in range(16, 25):
            rows = []
            for wv in range(8):
                a0, a1, a2, a3 = t[g, wv]
                nxt = t[g + 1, wv, 0]
                rows.append("w%d[%4d|%3d|%4d|%4d]" % (wv, a1 - a0, a2 - a1, a3 - a2, nxt - a3))
            print("  stage %2d (half1 | mid barrier | half2 | end barrier+epilogue): " % g + " ".join(rows))
        print("  stage starts relative to wave 0:", (t0[16:24] - t0[16:24, :1]).tolist())
        e = st.cpu().numpy().view(np.int64).reshape(-1)[3200:3200 + 12 * 8 * 8].reshape(12, 8, 8)
        for ti in (2, 3):
            base = t[ti * 8 + 7, :, 3]  # ts3 of the tile's last stage
            print("  tile %d epilogue, per wave [ts3->start | r0 transform | r0 barrier | r0 stores | r0 barrier+ | r1 transform | r1 barrier | r1 stores | ->next stage]:" % ti)
            for wv in range(8):
                v = e[ti, wv]
                nxt = t[ti * 8 + 8, wv, 0]
                print("    w%d" % wv, [int(v[0] - base[wv]), int(v[1] - v[0]), int(v[2] - v[1]), int(v[3] - v[2]), int(v[4] - v[3]), int(v[5] - v[4]), int(v[6] - v[5]), int(v[7] - v[6]), int(nxt - v[7])])
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
    # sustained regime: 150 back-to-back launches per figure after 50 warm-up launches (short bursts after an idle period run
    # up to 13 % faster or slower depending on the operand data: profiles/r02_conv_data_probe.txt)
    for mode in (0, 1):
        for _ in range(50): L.op_conv(x, w, b, 3, mode, sc, sh, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(150): L.op_conv(x, w, b, 3, mode, sc, sh, st)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 150)
    print("PIPE_ABL=%4d  mode0 %.3f ms  mode1 %.3f ms  (sustained)" % (v, res[0], res[1]), flush=True)
