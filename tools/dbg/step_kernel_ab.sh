# perf-debug: kernel trace of the fp32 step with ab/libssp_base.so (A) and the working tree (B), per-kernel differences (run through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/stepab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp --steps 10 --warmup 2 $*"
SSP_SKIP_ISA_VERIFY=1 SSP_HIP_LIB=$R/ab/libssp_base.so rocprofv3 --kernel-trace --output-format csv -d $O/a -o k -- python3 $R/bench.py $Q > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/b -o k -- python3 $R/bench.py $Q > /dev/null 2>&1
python3 $R/tools/dbg/step_kernel_diff.py $O/a $O/b > $O/diff.txt 2>&1
rm -rf $O/a $O/b
