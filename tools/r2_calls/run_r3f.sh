mkdir -p gpurun_out/r3f
timeout 600 python -m pytest tests/test_gpu_prepass.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r3f/pytest.txt
timeout 300 python tools/prepass_bench.py 2>&1 | tail -3 | tee gpurun_out/r3f/bench.txt
timeout 300 python tools/prepass_bench.py --shape 16,32,1024,128 2>&1 | tail -3 | tee -a gpurun_out/r3f/bench.txt
timeout 300 python tools/prepass_bench.py --shape 64,16,256,64 2>&1 | tail -3 | tee -a gpurun_out/r3f/bench.txt
timeout 300 python tools/prepass_trace.py --lib variants/libsage_gfx950_trace.so > gpurun_out/r3f/trace_c3.txt 2>&1; cat gpurun_out/r3f/trace_c3.txt
