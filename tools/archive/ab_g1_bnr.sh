R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/g1c; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  export SSP_G1_BNR=$v
  rocprofv3 --kernel-trace --stats -d $O/p$v -o n -- python3 $R/bench.py --no-cpu-baseline --traffic none --no-roofline --no-export --steps 6 --warmup 1 > /dev/null 2>&1
  find $O/p$v -name "*results.db" | head -1 | xargs -I{} python3 $R/tools/rocpd_stats.py {} 60 > $O/k$v.txt
  rm -rf $O/p$v
  echo "== BNR=$v"; grep -E "conv1x1|bn_bwd_kernel|wgrad_mfma" $O/k$v.txt | cut -c1-60,95-160
done
