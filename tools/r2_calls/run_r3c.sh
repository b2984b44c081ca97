mkdir -p gpurun_out/r3c
rm -f gpurun_out/r3c/prepass_slab.txt
timeout 600 python -m pytest tests/test_gpu_prepass.py -m gpu -x -q 2>&1 | tail -8 >> gpurun_out/r3c/prepass_slab.txt
for t in base noquant loadonly; do
  echo "== $t" >> gpurun_out/r3c/prepass_slab.txt
  timeout 300 python tools/prepass_bench.py --lib variants/libsage_gfx950_$t.so 2>&1 | tail -3 >> gpurun_out/r3c/prepass_slab.txt
done
timeout 300 python tools/prepass_bench.py --shape 16,32,1024,128 2>&1 | tail -3 >> gpurun_out/r3c/prepass_slab.txt
timeout 300 python tools/prepass_bench.py --shape 64,16,256,64 2>&1 | tail -3 >> gpurun_out/r3c/prepass_slab.txt
cat gpurun_out/r3c/prepass_slab.txt
