# perf-debug: same-box A/B of the bf16 pair step over values of one environment knob: ab_env.sh NAME v1 v2 ...  (run through gpurun)
N=$1; shift
Q="--dtype bf16 --no-cpu-baseline --traffic none --no-export --no-roofline --steps 40 --warmup 10"
for rep in 1 2; do for v in "$@"; do
  env $N=$v python bench.py $Q 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N=$v: %.1f pairs/s, %.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done > gpurun_out/ab_env.txt 2>&1
