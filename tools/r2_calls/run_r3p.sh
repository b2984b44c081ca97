mkdir -p gpurun_out/r3p
timeout 600 python -m pytest tests/test_gpu_prepass.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/prepass_bench.py > gpurun_out/r3p/prepass_c3.txt 2>&1; cat gpurun_out/r3p/prepass_c3.txt
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r3p/bench_c3.json; python -c "
import json; d=json.loads(open('gpurun_out/r3p/bench_c3.json').read()); print(d['value'], d['ms_per_step'], d['end_to_end'])"
