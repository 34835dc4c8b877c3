# perf-debug: phase trace of wgrad_wino_fused_kernel inside the fp32 pair step (run through gpurun).  Needs a TRACE build of the library:
#   SSP_HIPCC_EXTRA=-DWGF_TRACE=1 python -c "import __graft_entry__ as g; g.build()" ; cp semantic-superpoint_amd/csrc/libssp_hip.so ab/libssp_wgftrace.so
# (build it in the container, then rebuild the shipped library without the flag).  Output: gpurun_out/wgf_trace.txt
SSP_WGF_TRACE=${TRACE_EVERY:-7} SSP_SKIP_ISA_VERIFY=1 SSP_HIP_LIB=$PWD/ab/libssp_wgftrace.so python bench.py --no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp --steps 6 --warmup 2 2>&1 | grep -v "xcd\|amdgpu.ids" > gpurun_out/wgf_trace.txt
