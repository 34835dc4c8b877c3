R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/m8; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/p -o n -- python3 $R/bench.py --conv-algo 8 --no-cpu-baseline --traffic none --no-roofline --no-export --steps 6 --warmup 1 > $O/bench.json 2>/dev/null
find $O/p -name "*results.db" | head -1 | xargs -I{} python3 $R/tools/rocpd_stats.py {} 40 > $O/kernels.txt
rm -rf $O/p
cut -c1-75,95-150 $O/kernels.txt | head -28
