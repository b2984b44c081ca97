mkdir -p gpurun_out/r3d
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r3d/pytest_gpu.txt
timeout 300 python tools/prepass_bench.py > gpurun_out/r3d/prepass_c3.txt 2>&1; cat gpurun_out/r3d/prepass_c3.txt
timeout 300 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | tee gpurun_out/r3d/bench_c3.json
