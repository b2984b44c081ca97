mkdir -p gpurun_out/r3o
for t in base nt base nt; do
  echo "== $t" >> gpurun_out/r3o/nt.txt
  timeout 300 python tools/prepass_bench.py --lib variants/libsage_gfx950_$t.so 2>&1 | tail -2 | head -1 >> gpurun_out/r3o/nt.txt
  timeout 300 python tools/prepass_bench.py --lib variants/libsage_gfx950_$t.so --shape 16,32,1024,128 2>&1 | tail -2 | head -1 >> gpurun_out/r3o/nt.txt
done
cat gpurun_out/r3o/nt.txt
