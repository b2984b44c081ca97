mkdir -p gpurun_out/r3g
rm -f gpurun_out/r3g/sweep.txt
for shp in 2,32,8192,128 1,24,17550,128 4,32,4096,64 1,8,32768,128 8,32,2048,128 16,32,1024,128 32,16,512,128 64,16,256,64 2,4,8192,128 1,32,700,64 2,24,16384,128; do
  echo "== $shp" >> gpurun_out/r3g/sweep.txt
  timeout 300 python tools/prepass_bench.py --shape $shp 2>&1 | tail -3 >> gpurun_out/r3g/sweep.txt
done
cat gpurun_out/r3g/sweep.txt
