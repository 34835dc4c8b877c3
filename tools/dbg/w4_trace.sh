# perf-debug: phase trace of conv_wino4_kernel inside the fp32 pair step (run through gpurun).  Needs a TRACE build of the library:
#   SSP_HIPCC_EXTRA=-DW4_TRACE=1 python -c "import semantic_superpoint_amd as s; s.build(force=True)" ; cp .../libssp_hip.so ab/libssp_w4trace.so
# (build it in the container, then rebuild the shipped library without the flag).  Output: gpurun_out/w4_trace.txt
SSP_W4_TRACE=${TRACE_EVERY:-7} SSP_SKIP_ISA_VERIFY=1 SSP_HIP_LIB=$PWD/ab/libssp_w4trace.so python bench.py --no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp --steps 6 --warmup 2 2>&1 | grep -v "xcd\|amdgpu.ids" > gpurun_out/w4_trace.txt
