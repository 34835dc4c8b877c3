import torch, sys
sys.path.insert(0, '.')
from semantic_superpoint_amd import lib as L
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
N, H, W, cin, cout = 1, 16, 16, 64, 64
x = torch.randn(N, H, W, cin, generator=g).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, generator=g) / 24
a = L.op_conv_bf16(x.to(dev), w.to(dev), None, 3)
b = L.op_conv_bf16(x.float().to(dev), w.to(dev), None, 3)
torch.cuda.synchronize()
d = (a.float() - b.float()).abs()
print('max diff', d.max().item(), 'a absmax', a.float().abs().max().item(), 'b absmax', b.float().abs().max().item())
# which channels of the input matter: one-hot input channel tests
for ch in (0, 3, 4, 7, 8, 31, 32, 63):
    xx = torch.zeros(N, H, W, cin)
    xx[..., ch] = 1.0
    ww = torch.zeros(cout, cin, 3, 3)
    ww[:, :, 1, 1] = torch.arange(cin).float().view(1, -1) + 1
    o = L.op_conv_bf16(xx.to(dev), ww.to(dev), None, 3)
    torch.cuda.synchronize()
    print('in ch', ch, '-> out value', o[0, 5, 5, 0].item(), '(expect %d)' % (ch + 1))
