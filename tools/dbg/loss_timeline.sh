# perf-debug: timeline of the loss phase inside the bf16 pair step (run through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/losstl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python3 $R/bench.py --dtype bf16 --no-cpu-baseline --traffic none --no-export --no-roofline --steps 4 --warmup 2 > /dev/null 2>&1
python3 $R/tools/loss_phase_timeline.py $O/kt > $O/timeline.txt 2>&1
rm -rf $O/kt
