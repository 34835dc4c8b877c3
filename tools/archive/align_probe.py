"""Perf-debug: conv time versus the relative placement of the input and output tensors in HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from semantic_superpoint_amd import lib as L
dev = torch.device("cuda:0")
N, H, W, Cc = 32, 240, 320, 64
n_el = N * H * W * Cc
pool = torch.empty(3 * n_el + (64 << 20), dtype=torch.float32, device=dev)
base = pool.data_ptr()
w = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05
ws = torch.empty(4 * 16 * 16 * 64 * 4 + 1024, dtype=torch.uint8, device=dev)
lib = L.load_library()
def run(off_in_bytes, off_out_bytes):
    x = pool[off_in_bytes // 4: off_in_bytes // 4 + n_el]
    o = pool[off_out_bytes // 4: off_out_bytes // 4 + n_el]
    def call():
        L._check(lib.ssp_op_conv(L._ptr(x), L._ptr(w), None, L._ptr(o), N, H, W, Cc, Cc, 3, 0, None, None, None, 0,
                                 L._ptr(ws), ws.numel(), L._stream()))
    for _ in range(2): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); t = []
    for _ in range(7):
        e0.record(); call(); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
    return sorted(t)[3]
pool.normal_()
tensor_bytes = n_el * 4
print("base mod 2MiB = %d KiB" % ((base % (2 << 20)) >> 10))
for skew_kb in (0, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 1536, 2048 + 4, 3072, 4096 + 64):
    t = run(0, ((tensor_bytes + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + skew_kb * 1024)
    print("out = in + tensor (rounded to 2 MiB) + %5d KiB : %.3f ms" % (skew_kb, t))
