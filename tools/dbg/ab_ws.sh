# perf-debug: same-box A/B of the bf16 pair step: wave-specialised 3x3 kernels on / off  (run through gpurun)
Q="--dtype bf16 --no-cpu-baseline --traffic none --no-export --no-roofline --steps 40 --warmup 10"
for rep in 1 2; do for ws in 1 0; do
  SSP_CONVB_WS=$ws python bench.py $Q 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ws $ws: %.1f pairs/s, %.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done > gpurun_out/ab_ws.txt 2>&1
