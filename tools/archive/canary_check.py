"""Debug: run pair steps with every engine buffer embedded in a canary-filled arena; report out-of-bounds writes."""
import sys; sys.path.insert(0, '/root/repo')
import torch
from oracle import cpu_ref as C
from semantic_superpoint_amd.lib import Engine
dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "SuperPointNet_gauss2_ssmall"
B, H, W = 2, 64, 96
sd = C.init_state_dict(arch, seed=3)
ds = {k: v.to(dev).contiguous() for k, v in C.make_synthetic_pair(B, H, W, seed=8, semantic=arch.endswith("ssmall"), kp_prob=0.01).items()}
e = Engine(arch, B, H, W, dev)
GAP = 1 << 20
names = ["params", "grads", "adam_m", "adam_v", "bn_running", "nbt", "scalars", "workspace"]
sizes = {n: getattr(e, n).numel() * getattr(e, n).element_size() for n in names}
total = sum((s + 4095) // 4096 * 4096 + GAP for s in sizes.values()) + GAP + (8 << 20)
arena = torch.full((total,), 0xA5, dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + (4 << 20) - 1) // (4 << 20) * (4 << 20) - arena.data_ptr()  # 4 MiB aligned start
off = base + GAP
regions = {}
for n in names:
    t = getattr(e, n)
    view = arena[off:off + sizes[n]].view(t.dtype).view(t.shape)
    view.copy_(t)
    setattr(e, n, view)
    regions[n] = (off, sizes[n])
    off += (sizes[n] + 4095) // 4096 * 4096 + GAP
e.bind()
e.load_state_dict(sd)
torch.cuda.synchronize()
def check(tag):
    torch.cuda.synchronize()
    mask = torch.ones(total, dtype=torch.bool, device=dev)
    for n, (o, s) in regions.items():
        mask[o:o + s] = False
    bad = ((arena != 0xA5) & mask).nonzero().flatten()
    if bad.numel():
        b0 = int(bad[0]); b1 = int(bad[-1])
        near = min(regions.items(), key=lambda kv: min(abs(b0 - kv[1][0]), abs(b0 - kv[1][0] - kv[1][1])))
        print(tag, "OOB WRITES:", bad.numel(), "bytes; first at", b0, "last", b1, "nearest region", near[0], near[1], "delta from its end", b0 - near[1][0] - near[1][1])
        arena[bad] = 0xA5
    else:
        print(tag, "clean")
idx = e.sample_indices(ds["homographies"], 5)
check("sample_indices")
e.zero_grad(); e.pair_step(ds, indices=idx, train=True); check("pair_step given idx")
e.zero_grad(); e.pair_step(ds, indices=None, seed=3, train=True); check("pair_step sampled")
e.adam_step(0.001); check("adam")
e.zero_grad(); e.pair_step(ds, indices=idx, train=True, phase=1); e.pair_step(ds, indices=idx, train=True, phase=2); check("phases")
e.pair_step(ds, indices=idx, train=False); check("val step")
o = e.forward(ds["image"], slot=0, train=True, want=("semi", "desc")); check("forward")
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for s_ in (1, 2):
        e.zero_grad(); e.pair_step(ds, indices=None, seed=s_, train=True, graph=True)
torch.cuda.current_stream().wait_stream(st)
check("graph x2")
print("grad max", float(e.grads.abs().max()))
