#!/bin/bash
# The four headline lines + the bf16 kernel statistics at HEAD (a short form of refresh_profiles.sh; run through gpurun) -> gpurun_out/headline/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/headline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( time python3 $R/bench.py > $O/bench_default_run.json 2>/dev/null ) 2> $O/bench_default_run.time
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_ssp.json 2>/dev/null
python3 $R/bench.py --arch sp --steps 20 --warmup 5 --no-cpu-baseline --no-export > $O/bench_sp.json 2>/dev/null
python3 $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-export > $O/bench_ssp_bf16.json 2>/dev/null
python3 $R/bench.py --arch sp --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-export > $O/bench_sp_bf16.json 2>/dev/null
cd $R
bash tools/prof_bf16_quick.sh
cp gpurun_out/pbf16/kernel_stats.txt $O/bf16_kernel_stats.txt
cat $O/bench_default_run.time
