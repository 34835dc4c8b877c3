# perf-debug: samples rocm-smi (power, clocks) while an operator loop runs  (run through gpurun; $1 = conv | dgrad | wgrad)
W=${1:-dgrad}
python - <<PY &
import sys, time, torch
sys.path.insert(0, '.')
from semantic_superpoint_amd import lib as L
dev = torch.device('cuda:0')
N, H, Wd, cin, cout = 64, 240, 320, 64, 64
x = torch.randn(N, H, Wd, cin, device=dev).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, device=dev) / 24
sc = torch.rand(cin, device=dev) + 0.5
sh = torch.randn(cin, device=dev) * 0.3
dy = torch.randn(N, H, Wd, cout, device=dev).to(torch.bfloat16)
which = "$W"
def run():
    if which == "conv": return L.op_conv_bf16(x, w, None, 3, in_mode=1, in_scale=sc, in_shift=sh)
    if which == "dgrad": return L.op_conv_bf16(x, w, None, 3, in_mode=0)
    return L.op_conv_wgrad_bf16(x, dy, 3, in_mode=1, in_scale=sc, in_shift=sh)
for _ in range(5): run()
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(50): run()
    torch.cuda.synchronize(); n += 50
print("%s: %.3f ms per call over %d calls" % (which, (time.time() - t0) / n * 1e3, n))
PY
PID=$!
for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; sleep 0.3; kill -0 $PID 2>/dev/null || break; done | awk '{ if ($NF + 0 > 600) print }' | tail -3
wait $PID
