mkdir -p gpurun_out/r3e
timeout 300 python tools/prepass_trace.py --lib variants/libsage_gfx950_trace.so > gpurun_out/r3e/trace_c3.txt 2>&1; cat gpurun_out/r3e/trace_c3.txt
timeout 300 python tools/prepass_trace.py --lib variants/libsage_gfx950_trace.so --shape 64,16,256,64 > gpurun_out/r3e/trace_n256.txt 2>&1; cat gpurun_out/r3e/trace_n256.txt
