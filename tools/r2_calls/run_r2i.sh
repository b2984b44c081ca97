mkdir -p gpurun_out/r2i
for cfg in c3 n32k c5; do
  timeout 300 python tools/variant_bench.py --config $cfg --rounds 4 --reps 5 pipeb pipe old > gpurun_out/r2i/variants_$cfg.txt 2>&1
  grep -h "max|o\|median" gpurun_out/r2i/variants_$cfg.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config or kernel_vs_oracle or edge" 2>&1 | tail -3
