"""Debug: Winograd weight gradient vs torch on a few shapes; prints the relative error per 3x3 tap."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from semantic_superpoint_amd import lib as L
dev = torch.device("cuda:0")
L.set_conv_algo(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for (N, H, W, cin, cout, mode) in [(2, 16, 64, 64, 64, 0), (2, 16, 64, 64, 64, 1), (1, 36, 64, 64, 64, 1), (2, 30, 40, 128, 128, 1), (1, 32, 16, 64, 64, 0)]:
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.randn(N, H, W, cin).astype(np.float32)); dy = torch.from_numpy(rs.randn(N, H, W, cout).astype(np.float32))
    sc = torch.from_numpy(rs.uniform(-1.5, 1.5, cin).astype(np.float32)); sh = torch.from_numpy(rs.uniform(-0.5, 0.5, cin).astype(np.float32))
    xin = x if mode == 0 else torch.relu(x * sc + sh)
    ref = torch.nn.grad.conv2d_weight(xin.permute(0, 3, 1, 2).contiguous(), (cout, cin, 3, 3), dy.permute(0, 3, 1, 2).contiguous(), padding=1)
    out = L.op_conv_wgrad(x.to(dev), dy.to(dev), 3, mode, sc.to(dev), sh.to(dev)).cpu()
    e = (out - ref)
    print((N, H, W, cin, cout, mode), "rel %.3e" % (e.norm() / ref.norm()).item(),
          "per tap:", ["%.1e" % (e[:, :, i, j].norm() / ref[:, :, i, j].norm()).item() for i in range(3) for j in range(3)])
