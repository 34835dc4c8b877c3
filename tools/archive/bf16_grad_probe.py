"""Per-parameter gradient deviation of the reduced-precision conv modes from the fp32 step.
usage: python tools/bf16_grad_probe.py [B H W] [arch sp|ssp] [algos, default 3,7,8,0]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_superpoint_amd import lib as L, synth
dev = torch.device("cuda:0")
B, H, W = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 120, 160)
arch = "SuperPointNet_gauss2_ssmall" if (len(sys.argv) > 4 and sys.argv[4] == "ssp") else "SuperPointNet_gauss2"
algos = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "3,7,8,0").split(",")]
sample = synth.make_pair(B, H, W, dev, seed=3, semantic=arch.endswith("ssmall"))
sd = synth.default_init_state_dict(L.layer_table(arch), seed=0)
res = {}
e = L.Engine(arch, B, H, W, dev)
for a in [1] + algos:
    e.set_conv_algo(a)
    e.load_state_dict(sd)
    e.zero_grad()
    sc = e.pair_step(sample, seed=5, train=True)
    torch.cuda.synchronize()
    res[a] = ({k: v.clone().double() for k, v in e.grad_dict().items()}, sc.cpu().tolist()[:8])
print("scalars algo 1:", ["%.5f" % v for v in res[1][1]])
for a in algos:
    print("scalars algo %d:" % a, ["%.5f" % v for v in res[a][1]])
print("%-32s %10s " % ("parameter", "|g|") + " ".join("%12s" % ("algo %d relL2" % a) for a in algos))
worst = {a: 0.0 for a in algos}
for k in res[1][0]:
    g = res[1][0][k]
    if k.endswith(".bias") and ("conv" in k) and "Sout" not in k: continue
    n = float(g.norm())
    rel = {a: float((res[a][0][k] - g).norm()) / (n + 1e-30) for a in algos}
    for a in algos: worst[a] = max(worst[a], rel[a])
    print("%-32s %10.3e " % (k, n) + " ".join("%12.3e" % rel[a] for a in algos))
flat1 = torch.cat([v.reshape(-1) for v in res[1][0].values()])
for a in algos:
    fa = torch.cat([v.reshape(-1) for v in res[a][0].values()])
    print("algo %d: worst per-tensor rel-L2 %.3e, flat rel-L2 %.3e, cosine %.6f" % (a, worst[a], float((fa - flat1).norm() / flat1.norm()),
          float((fa * flat1).sum() / (fa.norm() * flat1.norm()))))
