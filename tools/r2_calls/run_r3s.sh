mkdir -p gpurun_out/r3s
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3s/prof -- python bench.py --no-cpu-baseline > gpurun_out/r3s/bench_under_rocprof.json 2> gpurun_out/r3s/rocprof.err
f=$(ls gpurun_out/r3s/prof/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/r3s/c3_kernel_stats.csv; head -5 gpurun_out/r3s/c3_kernel_stats.csv | cut -c1-200; rm -rf gpurun_out/r3s/prof
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r3s/bench_under_rocprof.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['end_to_end'])"
