"""Soak of the bf16 path (conv algo 12, BASELINE configs[3]) against fp32: two engines, identical initial weights,
identical batches and sampler seeds, N optimizer steps each; prints both loss curves (mean over 10-step windows) and
their relative deviation.  usage: soak_bf16.py [ssp|sp] [steps] [batch] [algo_a,algo_b]   (default 1,12; "9,1" soaks the default
algorithm - Winograd F(4x4,3x3) on the large maps - against F(2x2,3x3) only)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_superpoint_amd import synth, lib as L
from semantic_superpoint_amd.lib import Engine, layer_table, SCALAR_NAMES
arch = "SuperPointNet_gauss2" if (len(sys.argv) > 1 and sys.argv[1] == "sp") else "SuperPointNet_gauss2_ssmall"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
H, W = 240, 320
sd = synth.default_init_state_dict(layer_table(arch), seed=0)
engs = {}
pair = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "1,12").split(",")]
NAMES = {1: "fp32", 8: "mixed bf16 (algo 8)", 9: "fp32 F(2x2,3x3) only", 10: "fp32 F(4x4,3x3) forced", 12: "bf16 path (algo 12)"}
for name, algo in ((NAMES.get(pair[0], "algo %d" % pair[0]), pair[0]), (NAMES.get(pair[1], "algo %d" % pair[1]), pair[1])):
    L.set_conv_algo(algo)  # copied into the handle at creation
    engs[name] = Engine(arch, B, H, W, dev)
    engs[name].load_state_dict(sd)
L.set_conv_algo(1)
pool = [synth.make_pair(B, H, W, dev, seed=100 + k, semantic=arch.endswith("ssmall")) for k in range(4)]
li = SCALAR_NAMES.index("loss")
hist = {k: [] for k in engs}
for it in range(steps):
    for name, eng in engs.items():
        eng.zero_grad()
        eng.pair_step(pool[it % len(pool)], indices=None, seed=it + 1, train=True, lambda_loss=1.0, lamda_d=1.0, multi_task=True)
        eng.adam_step(1e-3)
        hist[name].append(eng.scalars[li].clone())
torch.cuda.synchronize()
na, nb = list(engs.keys())
f = torch.stack(hist[na]).cpu().double(); b = torch.stack(hist[nb]).cpu().double()
assert torch.isfinite(f).all() and torch.isfinite(b).all()
print("%s B=%d %dx%d, %d steps, 4 synthetic batches cycled, lr 1e-3; window means of the total loss" % (arch, B, H, W, steps))
print("%8s %24s %24s %10s" % ("steps", na, nb, "rel dev"))
worst = 0.0
for s0 in range(0, steps, 10):
    mf, mb = f[s0:s0 + 10].mean().item(), b[s0:s0 + 10].mean().item()
    dev_ = abs(mb - mf) / abs(mf)
    worst = max(worst, dev_)
    if s0 % 50 == 0 or s0 + 10 >= steps:
        print("%3d-%-4d %24.5f %24.5f %10.2e" % (s0, min(steps, s0 + 10) - 1, mf, mb, dev_))
print("per-step |b - a| / a: mean %.2e  max %.2e;   worst 10-step window %.2e" %
      (((b - f).abs() / f.abs()).mean().item(), ((b - f).abs() / f.abs()).max().item(), worst))
print("loss fell: %s %.4f -> %.4f, %s %.4f -> %.4f" % (na, f[0], f[-10:].mean(), nb, b[0], b[-10:].mean()))
