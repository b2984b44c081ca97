mkdir -p gpurun_out/r2b
for cfg in c3 n32k c5; do
  timeout 300 python tools/variant_bench.py --config $cfg --rounds 3 --reps 5 old magic lazy rot rotfold rot2w rotfold2w old2w > gpurun_out/r2b/variants_$cfg.txt 2>&1
done
timeout 120 python tools/variant_bench.py --config c5 --rounds 3 --reps 5 old rot rot4w > gpurun_out/r2b/variants_c5_4w.txt 2>&1
tail -12 gpurun_out/r2b/variants_c3.txt; tail -10 gpurun_out/r2b/variants_n32k.txt; tail -10 gpurun_out/r2b/variants_c5.txt; tail -4 gpurun_out/r2b/variants_c5_4w.txt
