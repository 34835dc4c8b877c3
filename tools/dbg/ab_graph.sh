# perf-debug: bf16 pair step eager vs captured hipGraph, same box  (run through gpurun)
Q="--dtype bf16 --no-cpu-baseline --traffic none --no-export --no-roofline --steps 40 --warmup 10"
for rep in 1 2; do for g in "" "--graph"; do
  python bench.py $Q $g 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 ${g:-eager}: %.1f pairs/s, %.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done > gpurun_out/ab_graph.txt 2>&1
