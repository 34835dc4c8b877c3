import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from semantic_superpoint_amd import lib as L
dev = torch.device("cuda:0")
N, H, W, C = 32, 240, 320, 64
x = torch.randn(N, H, W, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
b = torch.zeros(C, device=dev); sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
def run(tag, mode, stats, bias=True):
    st = torch.zeros(L.NREP, 2 * C, dtype=torch.float64, device=dev) if stats else None
    for _ in range(2): L.op_conv(x, w, b if bias else None, 3, mode, sc, sh, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); t = []
    for _ in range(7):
        e0.record(); L.op_conv(x, w, b if bias else None, 3, mode, sc, sh, st); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
    print("%-30s %.3f ms" % (tag, sorted(t)[3]))
run("mode0", 0, False); run("mode0 nobias", 0, False, False); run("mode0 stats", 0, True); run("mode1", 1, False); run("mode1 stats", 1, True)
