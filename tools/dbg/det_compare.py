"""perf-debug / validation: per-tensor gradient differences between deterministic and default accumulation (one step)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from semantic_superpoint_amd import lib as L, synth
ARCH = "SuperPointNet_gauss2_ssmall"
dev = torch.device("cuda:0")
B, H, W = 2, 120, 160
sd = synth.default_init_state_dict(L.layer_table(ARCH), seed=3)
sample = synth.make_pair(B, H, W, dev, seed=41, semantic=True)
def run(det):
    L.set_deterministic(det)
    e = L.Engine(ARCH, B, H, W, dev)
    e.load_state_dict(sd); e.zero_grad()
    sc = e.pair_step(sample, indices=None, seed=5, train=True).clone()
    torch.cuda.synchronize()
    g = {k: v.cpu().clone() for k, v in e.grad_dict().items()}
    L.set_deterministic(False)
    return sc.cpu(), g
runs = {"det1": run(True), "det2": run(True), "def1": run(False), "def2": run(False)}
def cmp(a, b):
    print("== %s vs %s: scalars max diff %.3e" % (a, b, float((runs[a][0] - runs[b][0]).abs().max())))
    rows = []
    for k in runs[a][1]:
        x, y = runs[a][1][k], runs[b][1][k]
        d = float((x - y).abs().max()); m = float(y.abs().max())
        rows.append((d / max(m, 1e-30), k, d, m))
    rows.sort(reverse=True)
    for r in rows[:3]: print("   %-40s max diff %.3e  (max |g| %.3e, ratio %.2e)" % (r[1], r[2], r[3], r[0]))
    rows.sort(key=lambda r: -r[2])
    print("   -- by absolute difference")
    for r in rows[:(60 if a != b and a[:3] != b[:3] else 4)]: print("   %-40s max diff %.3e  (max |g| %.3e, ratio %.2e)" % (r[1], r[2], r[3], r[0]))
cmp("det1", "det2"); cmp("def1", "def2"); cmp("det1", "def1")
