"""Achievable HBM bandwidth of this box with plain torch kernels (fill = write only, sum = read only, copy = both):
the yardstick for the HBM-bound kernels of the step."""
import torch
dev = torch.device("cuda:0")
n = 1258291200 // 4  # 1.26 GB, the size of one 240x320x64 activation of both views at B = 32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.0)); print("fill  (write 1.26 GB)        %.3f ms  %.2f TB/s" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: a.sum());      print("sum   (read 1.26 GB)         %.3f ms  %.2f TB/s" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: b.copy_(a));   print("copy  (read + write 2.52 GB) %.3f ms  %.2f TB/s" % (ms, 2 * n * 4 / ms / 1e9))
ms = t(lambda: torch.add(a, b, out=b)); print("add   (2 reads + write 3.77 GB) %.3f ms  %.2f TB/s" % (ms, 3 * n * 4 / ms / 1e9))
