# same-binary comparison of VALUES of an environment variable: tools/dbg/ab_envval.sh <tag> <VAR> "<v1> <v2> ..." <bench args...>
T=$1; V=$2; VALS=$3; shift; shift; shift
mkdir -p gpurun_out/$T
for i in 1 2; do
  for v in $VALS; do
    env $V=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-export --no-sp "$@" > gpurun_out/$T/v${v}_$i.json 2> gpurun_out/$T/v${v}_$i.err
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/$T/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f,"ERR",e)
PY
