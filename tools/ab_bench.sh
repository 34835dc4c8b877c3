#!/bin/bash
# Same-box A/B of two kernel builds: tools/ab_bench.sh <tag> [bench args]  ->  gpurun_out/<tag>/{new,base}{1,2}.json
# (SSP_HIP_LIB selects ab/libssp_base.so; alternating order so that clock drift shows up as a spread, not a bias)
T=$1; shift
mkdir -p gpurun_out/$T
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-export --no-sp "$@" > gpurun_out/$T/new$i.json 2> gpurun_out/$T/new$i.err
  SSP_SKIP_ISA_VERIFY=1 SSP_HIP_LIB=$PWD/ab/libssp_base.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-export --no-sp "$@" > gpurun_out/$T/base$i.json 2> gpurun_out/$T/base$i.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/$T/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f.split("/")[-1], d["value"], d["ms_per_step"], "conv avg ms", r.get("avg_launch_ms"), "exec frac", r.get("executed_frac"))
    except Exception as e: print(f, "ERR", e)
PY
