"""Per-parameter gradient deviation of the bf16-operand mode (algo 3) from the fp32 step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_superpoint_amd import lib as L, synth
dev = torch.device("cuda:0")
arch = "SuperPointNet_gauss2"
B, H, W = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 120, 160)
sample = synth.make_pair(B, H, W, dev, seed=3)
sd = synth.default_init_state_dict(L.layer_table(arch), seed=0)
res = {}
for a in (1, 3, 0):
    L.set_conv_algo(a)
    e = L.Engine(arch, B, H, W, dev)
    e.load_state_dict(sd)
    e.zero_grad()
    sc = e.pair_step(sample, seed=5, train=True)
    torch.cuda.synchronize()
    res[a] = ({k: v.clone().double() for k, v in e.grad_dict().items()}, sc.cpu().tolist()[:8])
L.set_conv_algo(1)
print("scalars fp32 :", ["%.5f" % v for v in res[1][1]])
print("scalars bf16 :", ["%.5f" % v for v in res[3][1]])
print("%-32s %10s %10s %10s" % ("parameter", "|g|", "bf16 relL2", "direct relL2"))
for k in res[1][0]:
    g = res[1][0][k]
    if k.endswith(".bias") and ("conv" in k) and "Sout" not in k: continue
    n = float(g.norm())
    print("%-32s %10.3e %10.3e %10.3e" % (k, n, float((res[3][0][k] - g).norm()) / (n + 1e-30), float((res[0][0][k] - g).norm()) / (n + 1e-30)))
