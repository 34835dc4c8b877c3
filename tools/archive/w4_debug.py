"""Debug helper: error pattern of the F(4x4,3x3) kernel on one small convolution."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from semantic_superpoint_amd import lib as L
dev = torch.device("cuda:0")
N, H, W, ci, co = 1, 16, 32, 64, 64
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(N, H, W, ci).astype(np.float32))
w = torch.from_numpy((rs.randn(co, ci, 3, 3) / np.sqrt(ci * 9)).astype(np.float32))
b = torch.zeros(co)
ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1).contiguous()
L.set_conv_algo(10)
out = L.op_conv(x.to(dev), w.to(dev), b.to(dev), 3, 0, None, None, None).cpu()
err = (out - ref).abs()
print("max err", float(err.max()), "ref max", float(ref.abs().max()))
bad = err > 1e-3
print("bad fraction", float(bad.float().mean()))
print("bad by row   ", bad.float().mean(dim=(0, 2, 3)).numpy().round(2))
print("bad by col   ", bad.float().mean(dim=(0, 1, 3)).numpy().round(2))
print("bad by chan  ", bad.float().mean(dim=(0, 1, 2)).numpy().round(2))
print("finite", bool(torch.isfinite(out).all()), "out sample", out[0, 0, 0, :4], ref[0, 0, 0, :4])
