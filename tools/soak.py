"""Sanity soak: N optimizer steps on one fixed synthetic pair batch; the loss must fall and stay finite."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_superpoint_amd import synth
from semantic_superpoint_amd.lib import Engine, layer_table, SCALAR_NAMES
arch = "SuperPointNet_gauss2_ssmall" if (len(sys.argv) > 1 and sys.argv[1] == "ssp") else "SuperPointNet_gauss2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
B, H, W = 16, 240, 320
eng = Engine(arch, B, H, W, dev)
eng.load_state_dict(synth.default_init_state_dict(layer_table(arch), seed=0))
sample = synth.make_pair(B, H, W, dev, seed=7, semantic=arch.endswith("ssmall"))
first = last = None
for it in range(steps):
    eng.zero_grad()
    eng.pair_step(sample, indices=None, seed=it + 1, train=True, lambda_loss=1.0, lamda_d=1.0, multi_task=True)
    eng.adam_step(1e-3)
    if it % 50 == 0 or it == steps - 1:
        s = dict(zip(SCALAR_NAMES, eng.scalars.cpu().tolist()))
        assert all(v == v and abs(v) < 1e6 for v in s.values()), s
        print(it, {k: round(v, 4) for k, v in s.items() if k.startswith("loss") or k.endswith("dist")}, flush=True)
        first = first or s["loss"]; last = s["loss"]
assert last < first, (first, last)
print("soak ok: loss %.4f -> %.4f" % (first, last))
