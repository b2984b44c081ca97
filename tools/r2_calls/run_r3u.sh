mkdir -p gpurun_out/r3u
timeout 200 python tools/shape_probe.py > gpurun_out/r3u/shape_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for grp in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU"; do
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/r3u/p -- python tools/run_prepass.py fused 2,32,8192,128 3 > gpurun_out/r3u/p.log 2>&1
  f=$(ls gpurun_out/r3u/p/*/*counter_collection.csv 2>/dev/null | head -1)
  python3 - "$f" >> gpurun_out/r3u/pmc_prepass_valu.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "prepass_kv_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):.5e}   (n={len(v)})")
PY
  rm -rf gpurun_out/r3u/p
done
cat gpurun_out/r3u/pmc_prepass_valu.txt; tail -16 gpurun_out/r3u/shape_probe.txt
