# perf-debug: conv_bf16_ws_kernel phase trace of launches INSIDE the pair step (every 29th 3x3 launch)  (run through gpurun)
SSP_CONVB_TRACE=${TRACE_EVERY:-29} python bench.py --dtype bf16 --no-cpu-baseline --traffic none --no-export --no-roofline --steps 30 --warmup 5 2>&1 | grep -v "xcd\|amdgpu.ids" > gpurun_out/instep_trace.txt
