#!/usr/bin/env python3
"""correctness probe (run through gpurun): the sequence of tests/test_gpu_bf16_path.py::test_bf16_apply_pass_fused_... in one process
(three engines one after the other, inputs re-uploaded per step, SSP_BF16_FUSE_APPLY 1, 0, 1, 1) with a report of WHICH step is the odd one."""
import os, sys, gc
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
# (the GPU's unique id BEFORE anything initialises the GPU: no child processes afterwards)
import subprocess
try:
    uid = subprocess.run(["rocm-smi", "--showuniqueid"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20).stdout
    uid = " ".join(l.split(":")[-1].strip() for l in uid.splitlines() if "Unique ID" in l)
except Exception as ex:
    uid = "?"
import torch
from oracle import cpu_ref as C
from semantic_superpoint_amd import lib as L
dev = torch.device("cuda:0")
import socket
pr = torch.cuda.get_device_properties(0)
print("box %s gpu %s: %s, %d CUs, %.0f GB" % (socket.gethostname(), uid, pr.name, pr.multi_processor_count, pr.total_memory / 2**30), flush=True)
def to_dev(s): return {k: v.to(dev).contiguous() for k, v in s.items()}
for (arch, B, H, W) in [("SuperPointNet_gauss2_ssmall", 2, 120, 160), ("SuperPointNet_gauss2", 2, 240, 320), ("SuperPointNet_gauss2", 1, 72, 104)]:
    sd = C.init_state_dict(arch, seed=12)
    sample = C.make_synthetic_pair(B, H, W, seed=6, semantic=arch.endswith("ssmall"), kp_prob=0.005)
    L.set_deterministic(True)
    e = L.Engine(arch, B, H, W, dev)
    e.set_conv_algo(12)
    e.load_state_dict(sd)
    idx = e.sample_indices(to_dev(sample)["homographies"], seed=5)
    out = []
    for mode in ("1", "0", "1", "1", "0"):
        os.environ["SSP_BF16_FUSE_APPLY"] = mode
        e.zero_grad()
        sc = e.pair_step(to_dev(sample), indices=idx, train=True)
        torch.cuda.synchronize()
        out.append((sc.cpu().clone(), {k: v.cpu().clone() for k, v in e.grad_dict().items()}))
    del e
    L.set_deterministic(False)
    def nbad(i, j): return len([k for k in out[i][1] if not torch.equal(out[i][1][k], out[j][1][k])])
    print("%s %dx%d: fused steps 0|2 differ in %d tensors, 0|3: %d, 2|3: %d; separate steps 1|4: %d; scalars 0==2 %s" % (
        arch, H, W, nbad(0, 2), nbad(0, 3), nbad(2, 3), nbad(1, 4), torch.equal(out[0][0], out[2][0])), flush=True)
    for (i, j) in ((0, 2), (1, 4)):
        bad = [k for k in out[i][1] if not torch.equal(out[i][1][k], out[j][1][k])]
        for k in bad[:40]:
            a, b = out[i][1][k].double(), out[j][1][k].double()
            print("    steps %d|%d %s rel-L2 %.3e" % (i, j, k, float((a - b).norm() / (b.norm() + 1e-30))))
