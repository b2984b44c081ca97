mkdir -p gpurun_out/r3q
for t in nt1 nt0 nt1 nt0; do
  SAGE_GFX950_LIB=$PWD/variants/libsage_gfx950_$t.so timeout 300 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t', 'kernel', d['ms_per_step'], 'e2e', d['end_to_end']['ms_per_call'], 'prepass', d['end_to_end']['prepass']['avg_launch_ms'])" | tee -a gpurun_out/r3q/e2e_nt.txt
done
