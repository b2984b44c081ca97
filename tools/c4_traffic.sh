#!/bin/bash
# C4 causal: does the launch's time follow its L2-miss traffic?  The packed launch's plan deals the query heads to the XCDs in groups; forcing
# the group size (SAGE_ORDER_GROUP = heads whose items interleave on an XCD; automatic = 4 = one GQA group = one K/V stream per XCD) changes how
# many K/V streams share an XCD's 4 MB L2, i.e. the traffic, without changing the work.  Per setting: the kernel's time (HIP events) and
# FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, a pass each).
out=gpurun_out/${TAG:-r5d}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in -1 1 2 8 16; do
  export SAGE_ORDER_GROUP=$g
  t=$(timeout 200 python tools/lib_ab.py c4 main 2>&1 | grep "median" | sed 's/.*median *\([0-9.]*\) us.*/\1/')
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/p; timeout 150 rocprofv3 --pmc $ctr --output-format csv -d $out/p -- python tools/run_kernel.py c4 3 > $out/p.log 2>&1
    eval "$ctr=$(python3 - $out/p $ctr <<'PY'
import csv, glob, sys
v = [float(r["Counter_Value"]) for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv") for r in csv.DictReader(open(f))
     if "sage_attn_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
print(sum(v) / max(len(v), 1))
PY
)"
  done
  python3 -c "print('group %3s   kernel %8.1f us   FETCH %9.0f KiB  WRITE %9.0f KiB   HBM-side %.1f MB (x %.2f algorithmic)' % ('$g', $t, $FETCH_SIZE, $WRITE_SIZE, (2*$FETCH_SIZE+$WRITE_SIZE)*1024/1e6, (2*$FETCH_SIZE+$WRITE_SIZE)*1024/651.9e6))" | tee -a $out/c4_traffic.txt
done
rm -rf $out/p $out/p.log
