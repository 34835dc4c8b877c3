# perf-debug: fp32 and bf16 pair-step rates of the current build (run through gpurun)
Q="--no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp --steps 40 --warmup 10"
for rep in 1 2; do for dt in f32 bf16; do
  python bench.py --dtype $dt $Q 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt: %.1f pairs/s, %.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done > gpurun_out/ab_quick.txt 2>&1
