"""Times conv_bf16_kernel / wgrad_bf16_kernel on the layer shapes of the benchmark (operator level)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from semantic_superpoint_amd import lib as L
dev = torch.device('cuda:0')
shapes = [(64, 240, 320, 64, 64), (64, 120, 160, 64, 64), (64, 60, 80, 128, 128), (64, 30, 40, 128, 128), (64, 30, 40, 128, 256)]
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
for (N, H, W, cin, cout) in shapes:
    x = torch.randn(N, H, W, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) / 24
    sc = torch.rand(cin, device=dev) + 0.5
    sh = torch.randn(cin, device=dev) * 0.3
    dy = torch.randn(N, H, W, cout, device=dev).to(torch.bfloat16)
    def run():
        if which == "conv":
            return L.op_conv_bf16(x, w, None, 3, in_mode=1, in_scale=sc, in_shift=sh)
        if which == "dgrad":
            return L.op_conv_bf16(x, w, None, 3, in_mode=0)
        return L.op_conv_wgrad_bf16(x, dy, 3, in_mode=1, in_scale=sc, in_shift=sh)
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 10
    for _ in range(n): run()
    torch.cuda.synchronize()
    ms = (time.time() - t0) / n * 1e3
    fl = 2.0 * N * H * W * cin * cout * 9
    by = 2.0 * N * H * W * (cin + cout)
    print("%s %s: %.3f ms  %.0f TF/s  %.2f TB/s (algorithmic)" % (which, (N, H, W, cin, cout), ms, fl / ms / 1e9, by / ms / 1e9))
