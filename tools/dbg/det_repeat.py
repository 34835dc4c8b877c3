#!/usr/bin/env python3
"""perf-debug / correctness probe (run through gpurun): N identical pair steps of ONE fresh engine under ssp_set_deterministic and
prints, per step, which gradient tensors differ from step 0 / from the previous step (bit comparison) and the scalars.
usage: python tools/dbg/det_repeat.py [arch=sp|ssp] [H W] [algo=12] [steps=5] [fuse modes per step, e.g. 10]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from oracle import cpu_ref as C
from semantic_superpoint_amd import lib as L
arch = {"sp": "SuperPointNet_gauss2", "ssp": "SuperPointNet_gauss2_ssmall"}[sys.argv[1] if len(sys.argv) > 1 else "sp"]
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (240, 320)
algo = int(sys.argv[4]) if len(sys.argv) > 4 else 12
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
modes = sys.argv[6] if len(sys.argv) > 6 else "1"   # SSP_BF16_FUSE_APPLY per step, cycled ("101": fused, separate, fused, ...)
dev = torch.device("cuda:0")
B = 2
sd = C.init_state_dict(arch, seed=12)
sample = {k: v.to(dev).contiguous() for k, v in C.make_synthetic_pair(B, H, W, seed=6, semantic=arch.endswith("ssmall"), kp_prob=0.005).items()}
L.set_deterministic(True)
e = L.Engine(arch, B, H, W, dev)
e.set_conv_algo(algo)
e.load_state_dict(sd)
idx = e.sample_indices(sample["homographies"], seed=5)
outs = []
for s in range(steps):
    os.environ["SSP_BF16_FUSE_APPLY"] = modes[s % len(modes)]
    e.zero_grad()
    sc = e.pair_step(sample, indices=idx, train=True)
    torch.cuda.synchronize()
    outs.append((sc.cpu().clone(), {k: v.cpu().clone() for k, v in e.grad_dict().items()}))
for s in range(1, steps):
    if modes[s % len(modes)] != modes[0]:
        continue
    bad0 = [k for k, g in outs[s][1].items() if not torch.equal(g, outs[0][1][k])]
    badp = [k for k, g in outs[s][1].items() if not torch.equal(g, outs[s - 1][1][k])]
    print("step %d: scalars equal step 0: %s | %d tensors differ from step 0, %d from step %d" % (
        s, torch.equal(outs[s][0], outs[0][0]), len(bad0), len(badp), s - 1), bad0[:4], flush=True)
    for k in bad0[:3]:
        a, b = outs[s][1][k].double(), outs[0][1][k].double()
        print("    %s rel-L2 %.3e" % (k, float((a - b).norm() / (b.norm() + 1e-30))))
