mkdir -p gpurun_out/r3b
for t in base nowait noquant nostats loadonly; do
  echo "== $t" >> gpurun_out/r3b/prepass_abl.txt
  timeout 300 python tools/prepass_bench.py --lib variants/libsage_gfx950_$t.so 2>&1 | tail -3 >> gpurun_out/r3b/prepass_abl.txt
done
cat gpurun_out/r3b/prepass_abl.txt
