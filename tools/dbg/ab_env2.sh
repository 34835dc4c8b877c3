# perf-debug: same-box A/B of the pair step (fp32 and bf16) over values of one environment knob: ab_env2.sh NAME v1 v2 ...
N=$1; shift
Q="--no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp --steps 40 --warmup 10"
for rep in 1 2; do for v in "$@"; do for dt in f32 bf16; do
  env $N=$v python bench.py --dtype $dt $Q 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N=$v $dt: %.1f pairs/s, %.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done; done > gpurun_out/ab_env2.txt 2>&1
