"""Debug helper (GPU box): per-parameter gradient error of ssp_backward vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cpu_ref as C
from semantic_superpoint_amd.lib import Engine

arch = sys.argv[1] if len(sys.argv) > 1 else "SuperPointNet_gauss2"
which = sys.argv[2] if len(sys.argv) > 2 else "all"
B, H, W = 2, 64, 96
sd = C.init_state_dict(arch, seed=5)
rs = np.random.RandomState(77)
x = torch.from_numpy(rs.uniform(0, 1, (B, 1, H, W)).astype(np.float32))
tsd = C.to_torch(sd, requires_grad=True)
ref = C.forward(tsd, x, arch)
gs = {k: torch.from_numpy(rs.randn(*ref[k].shape).astype(np.float32)) for k in ref}
if "sem" in gs:
    gs["sem"] *= 0.05
if which != "all":
    for k in gs:
        if k != which:
            gs[k] = None
loss = sum((ref[k] * gs[k]).sum() for k in ref if gs[k] is not None)
loss.backward()
dev = torch.device("cuda:0")
e = Engine(arch, B, H, W, dev)
e.load_state_dict(sd)
e.forward(x.to(dev), slot=0, train=True, want=())
e.zero_grad()
f = lambda t: None if t is None else t.to(dev)
e.backward(0, f(gs["semi"]), f(gs["desc"]), f(gs.get("sem")))
torch.cuda.synchronize()
gd = e.grad_dict()
for k in C.param_keys(arch):
    r = tsd[k].grad
    if r is None:
        r = torch.zeros_like(tsd[k])
    m = gd[k].cpu()
    print("%-34s ref_max %.4e  mine_max %.4e  err %.4e  rel %.3e" % (k, r.abs().max(), m.abs().max(), (m - r).abs().max(),
          (m - r).abs().max() / (r.abs().max() + 1e-12)))
