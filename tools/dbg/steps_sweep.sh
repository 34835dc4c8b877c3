# perf-debug: does the measured rate depend on how long the run is (clock ramp of the power controller)?  (run through gpurun)
Q="--no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp"
for dt in f32 bf16; do for k in 20 100 400; do
  python bench.py --dtype $dt --steps $k --warmup 5 $Q 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt steps $k: %.1f pairs/s, %.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done > gpurun_out/steps_sweep.txt 2>&1
python bench.py --dtype bf16 --steps 20 --warmup 100 $Q 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 steps 20 warmup 100: %.1f pairs/s, %.3f ms/step' % (d['value'], d['ms_per_step']))" >> gpurun_out/steps_sweep.txt 2>&1
