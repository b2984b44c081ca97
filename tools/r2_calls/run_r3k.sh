mkdir -p gpurun_out/r3k
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r3k/pytest_gpu.txt
timeout 300 python bench.py --config c2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r3k/bench_c2.json
