# perf-debug: what deterministic mode costs, per kernel (run through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/detcost; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp"
for d in 0 1; do
  SSP_DETERMINISTIC=$d rocprofv3 --kernel-trace --stats -d $O/kt$d -o k -- python3 $R/bench.py --dtype bf16 $Q --steps 6 --warmup 1 > $O/line$d.txt 2>/dev/null
  find $O/kt$d -name "*results.db" | head -1 | xargs -I{} python3 $R/tools/rocpd_stats.py {} 40 > $O/stats$d.txt
  rm -rf $O/kt$d
done
