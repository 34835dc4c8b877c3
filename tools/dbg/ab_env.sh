# same-binary A/B of an environment switch: tools/dbg/ab_env.sh <tag> <VAR> <bench args...>  (VAR=1 vs VAR=0, both orders)
T=$1; V=$2; shift; shift
mkdir -p gpurun_out/$T
for i in 1 2; do
  env $V=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-export --no-sp "$@" > gpurun_out/$T/on$i.json 2> gpurun_out/$T/on$i.err
  env $V=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-export --no-sp "$@" > gpurun_out/$T/off$i.json 2> gpurun_out/$T/off$i.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/$T/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f,"ERR",e)
PY
