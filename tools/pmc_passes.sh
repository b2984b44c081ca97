#!/bin/bash
# PMC passes over the attention kernel only (tools/run_kernel.py), one rocprofv3 --pmc run per counter group
# (no trace domains combined with --pmc).  Prints the mean per-dispatch value of each counter for sage_attn_kernel.
# config: env SAGE_PMC_CFG (default c3)
# usage: tools/pmc_passes.sh <outdir> "<group1 counters>" "<group2 counters>" ...
set -u
out="$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- python tools/run_kernel.py ${SAGE_PMC_CFG:-c3} 3 > "$out/p$i.log" 2>&1
  f=$(ls "$out"/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "pass $i: no counter file"; tail -3 "$out/p$i.log"; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "sage_attn" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} {sum(v)/len(v):.5e}   (n={len(v)})")
PY
  rm -rf "$out/p$i"
done
