mkdir -p gpurun_out/r3i
timeout 900 python -m pytest tests/test_gpu_prepass.py -m gpu -x -q --timeout 300 2>&1 | tail -5 | tee gpurun_out/r3i/pytest.txt
timeout 300 python tools/prepass_bench.py 2>&1 | tail -3 | tee gpurun_out/r3i/bench.txt
