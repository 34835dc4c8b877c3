"""bf16 path (algo 12) vs the fp32 default on one pair step: scalars and per-tensor gradient agreement."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import cpu_ref as C
from semantic_superpoint_amd.lib import Engine, SCALAR_NAMES
dev = torch.device('cuda:0')
arch = sys.argv[1] if len(sys.argv) > 1 else "SuperPointNet_gauss2_ssmall"
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 120, 160)
sd = C.init_state_dict(arch, seed=1)
sample = C.make_synthetic_pair(B, H, W, seed=2, semantic=arch.endswith("ssmall"))
sample = {k: v.to(dev).contiguous() for k, v in sample.items()}
res = {}
for algo in (1, 12):
    eng = Engine(arch, B, H, W, dev)
    eng.set_conv_algo(algo)
    eng.load_state_dict(sd)
    eng.zero_grad()
    idx = eng.sample_indices(sample["homographies"], 7)
    sc = eng.pair_step(sample, indices=idx, train=True)
    torch.cuda.synchronize()
    res[algo] = (dict(zip(SCALAR_NAMES, sc.cpu().tolist())), {k: v.clone().cpu() for k, v in eng.grad_dict().items()})
    t0 = time.time()
    for _ in range(5):
        eng.zero_grad(); eng.pair_step(sample, indices=idx, train=True)
    torch.cuda.synchronize()
    print("algo", algo, "ms/step", (time.time() - t0) / 5 * 1e3)
    del eng
a, b = res[1], res[12]
for k in a[0]:
    print("%-16s fp32 %.6f  bf16 %.6f" % (k, a[0][k], b[0][k]))
worst = 0
for k in a[1]:
    ga, gb = a[1][k].double().flatten(), b[1][k].double().flatten()
    rel = float((ga - gb).norm() / (ga.norm() + 1e-30))
    cos = float((ga @ gb) / (ga.norm() * gb.norm() + 1e-30))
    worst = max(worst, rel)
    print("%-40s |g| %.3e rel-L2 %.3e cos %.6f" % (k, float(ga.norm()), rel, cos))
print("worst rel", worst)
