# perf-debug: conv_wino4_kernel phase traces of two TRACE builds on the same box (ab/libssp_w4trace.so = working tree,
# ab/libssp_w4trace_base.so = a base revision; both built with -DW4_TRACE=1, see w4_trace.sh) -> gpurun_out/w4_trace_{new,base}.txt
for v in new base; do
  L=$PWD/ab/libssp_w4trace.so; [ $v = base ] && L=$PWD/ab/libssp_w4trace_base.so
  SSP_W4_TRACE=${TRACE_EVERY:-7} SSP_SKIP_ISA_VERIFY=1 SSP_HIP_LIB=$L python bench.py --no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp --steps 6 --warmup 2 2>&1 | grep -v "xcd\|amdgpu.ids" > gpurun_out/w4_trace_$v.txt
done
