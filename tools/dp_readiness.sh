#!/bin/bash
# Multi-GPU readiness on a 1-GPU box: two ranks of bench.py on cuda:0 over gloo (SSP_BENCH_SINGLE_DEVICE=1),
# with and without the overlapped split-bucket all-reduce, and a rocprofv3 kernel + memory-copy trace of rank 0 of the overlapped
# run.  Every rank is started by THIS shell (bench.py's own spawner would fork from a profiled process).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/dp_readiness
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SSP_BENCH_SINGLE_DEVICE=1 MASTER_ADDR=127.0.0.1 WORLD_SIZE=2 LOCAL_WORLD_SIZE=2
ARGS="--gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --traffic none --no-export --no-roofline --batch 16"
run2() {  # $1 = tag, $2 = extra args, $3 = port, $4 = profile rank 0 (0/1)
  RANK=1 LOCAL_RANK=1 MASTER_PORT=$3 python3 $R/bench.py $ARGS $2 > /dev/null 2> $O/$1.rank1.err &
  P1=$!
  if [ "$4" = 1 ]; then
    RANK=0 LOCAL_RANK=0 MASTER_PORT=$3 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_$1 -o t -- python3 $R/bench.py $ARGS $2 > $O/$1.json 2> $O/$1.rank0.err
  else
    RANK=0 LOCAL_RANK=0 MASTER_PORT=$3 python3 $R/bench.py $ARGS $2 > $O/$1.json 2> $O/$1.rank0.err
  fi
  wait $P1
}
run2 overlap "" 29611 0
run2 no_overlap "--no-overlap" 29612 0
run2 overlap_traced "" 29613 1
run2 overlap_bf16 "--dtype bf16" 29614 0
python3 - <<PY
import json
for t in ("overlap", "no_overlap", "overlap_traced", "overlap_bf16"):
    try:
        d = json.loads(open("$O/%s.json" % t).read().strip().splitlines()[-1])
        print(t, d["value"], "pairs/s", d["ms_per_step"], "ms/step", d["config"]["allreduce"])
        print("   dp:", json.dumps(d.get("dp"))[:1200])
    except Exception as e:
        print(t, "ERR", e)
PY
python3 $R/tools/archive/overlap_timeline.py $O/trace_overlap_traced > $O/timeline.txt 2>&1
cat $O/timeline.txt
rm -rf $O/trace_overlap_traced
