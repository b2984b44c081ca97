#!/bin/bash
# HBM-side traffic of the K / V pre-pass from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, KiB per dispatch,
# summed over the kernels of one pre-pass).  The load-only build (-DSAGE_PP_ABL=7) reads a known 268.4 MB with the same
# 8-byte-per-lane buffer loads: it calibrates FETCH_SIZE for this access width (the guide's x2 rule is for 16 B per lane).
# usage: tools/pmc_prepass.sh <outdir>
set -u
out="$1"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # tag mode extra-args
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 180 rocprofv3 --pmc $ctr --output-format csv -d "$out/p" -- python tools/run_prepass.py $2 2,32,8192,128 3 $3 > "$out/p.log" 2>&1
    f=$(ls "$out"/p/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -z "$f" ] && { echo "$1 $ctr: no counter file"; tail -3 "$out/p.log"; continue; }
    python3 - "$f" "$1" "$ctr" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "sage::" in n and r["Counter_Name"] == sys.argv[3]:
        acc[n.split("<")[0].split("(")[0]].append(float(r["Counter_Value"]))
tot = 0.0
for k, v in sorted(acc.items()):
    per = sum(v) / 3.0                      # 3 repetitions of the pre-pass
    tot += per
    print(f"{sys.argv[2]:10s} {sys.argv[3]:10s} {k:40s} {per:12.0f} KiB per pre-pass ({len(v)} dispatches)")
print(f"{sys.argv[2]:10s} {sys.argv[3]:10s} {'TOTAL':40s} {tot:12.0f} KiB = {tot*1024/1e6:8.1f} MB")
PY
    rm -rf "$out/p"
  done
}
run loadonly fused "--lib variants/libsage_gfx950_loadonly.so"
run fused fused ""
run sequence sequence ""
