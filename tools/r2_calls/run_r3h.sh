mkdir -p gpurun_out/r3h
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r3h/pytest_gpu.txt
timeout 300 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | tee gpurun_out/r3h/bench_c3.json
timeout 300 python bench.py --config c5 --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/r3h/bench_c5.json
timeout 300 python tools/shape_probe.py > gpurun_out/r3h/shape_probe.txt 2>&1; cat gpurun_out/r3h/shape_probe.txt
