#!/bin/bash
# Samples GPU power / clocks (rocm-smi) while bench.py runs: is the fp32 pair step power-limited?
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 3000 --warmup 3 "$@" > /tmp/pp_bench.json 2>/dev/null &
BP=$!
sleep 30
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i -E "power|sclk|mclk|junction|edge" | tr '\n' ';' ; echo
  sleep 0.7
done
wait $BP
cut -c60-140 /tmp/pp_bench.json
