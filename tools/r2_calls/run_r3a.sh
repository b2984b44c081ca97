mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_prepass.py -m gpu -x -q 2>&1 | tail -15
timeout 300 python tools/prepass_bench.py > gpurun_out/r3a/prepass_c3.txt 2>&1; cat gpurun_out/r3a/prepass_c3.txt
timeout 300 python tools/prepass_bench.py --shape 16,32,1024,128 > gpurun_out/r3a/prepass_n1024.txt 2>&1; tail -4 gpurun_out/r3a/prepass_n1024.txt
timeout 300 python tools/prepass_bench.py --shape 64,16,256,64 > gpurun_out/r3a/prepass_n256.txt 2>&1; tail -4 gpurun_out/r3a/prepass_n256.txt
