"""perf-debug: dumps the raw bf16 output of layer 1 (and its pooled copy) of a 240x320 forward to gpurun_out/<tag>_y1.pt (A/B of two libraries)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import cpu_ref as C
from semantic_superpoint_amd.lib import Engine
tag = sys.argv[1]
arch, B, H, W = "SuperPointNet_gauss2", 2, 240, 320
sd = C.init_state_dict(arch, seed=5)
x = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(3))
e = Engine(arch, B, H, W, torch.device("cuda:0"), with_grad=False)
e.set_conv_algo(12)
e.load_state_dict(sd)
e.forward(x.cuda(), slot=0, train=True, want=("semi", "desc"))
torch.cuda.synchronize()
y1 = e.debug_buffer(0, "Y1", (B, H, W, 64), torch.bfloat16).cpu().float()
a1 = e.debug_buffer(0, "A1", (B, H // 2, W // 2, 64), torch.bfloat16).cpu().float()
torch.save({"y1": y1, "a1": a1}, "gpurun_out/%s_y1.pt" % tag)
print(tag, float(y1.abs().sum()), float(a1.abs().sum()))
