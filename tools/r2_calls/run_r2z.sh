mkdir -p gpurun_out/r2z
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_kv" 2>&1 | tail -12
timeout 300 python tools/shape_probe.py > gpurun_out/r2z/shape_probe.txt 2>&1; cat gpurun_out/r2z/shape_probe.txt
