"""Segmentation-loss operator alone at the training shape (32 x 240x320, 133 classes): both lane layouts, segment-like and
per-pixel-random labels.  usage: python tools/dbg/sem_time.py [reps]"""
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from semantic_superpoint_amd import lib as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
B, Hc, Wc, C, cs = 32, 30, 40, 133, 136
gen = torch.Generator().manual_seed(0)
x = torch.zeros(B, Hc, Wc, cs, device=dev)
x[..., :C] = (torch.randn(B, Hc, Wc, C, generator=gen) * 3).to(dev)
d = torch.empty_like(x)
out = torch.zeros(1, device=dev)
scratch = torch.empty(65536, dtype=torch.uint8, device=dev)
lib = L.load_library()
for kind in ("segments", "noise"):
    if kind == "noise":
        lab = torch.randint(0, C, (B, 8 * Hc, 8 * Wc), generator=gen)
    else:
        coarse = torch.randint(0, C, (B, 1, 10, 14), generator=gen).float()
        lab = F.interpolate(coarse, size=(8 * Hc, 8 * Wc), mode="nearest")[:, 0].long()
    lab = lab.to(dev)
    res = {}
    for algo in (1, 2, 1, 2):
        for train in (True, False):
            def run():
                L._check(lib.ssp_op_sem_loss(L._ptr(x), cs, L._ptr(lab), B, 8 * Hc, 8 * Wc, C, algo, L._ptr(scratch), scratch.numel(),
                                             L._ptr(out), L._ptr(d) if train else None, L._stream()))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[(algo, train)] = (e0.elapsed_time(e1) / reps * 1e3, float(out.item()), float(d.abs().sum().item()) if train else 0.0)
    for k, v in sorted(res.items()):
        print("%-9s algo %d %-8s %8.1f us per call (count + zero + loss kernel + finish)   loss %.6f  sum|d| %.6f"
              % (kind, k[0], "train" if k[1] else "forward", v[0], v[1], v[2]))
