#!/bin/bash
# A/B of the causal work order (variants built by tools/build_variants.sh: ord0 = head-major, ordA = default grouping, ordN = groups of N)
for c in ${CONFIGS:-n1k n2k n4k c3 n32k c2 d64f8}; do
  echo "== $c"
  timeout 200 python tools/variant_bench.py --config $c --rounds 5 --reps 10 "$@" 2>&1 | grep -v "^{" | tail -n +1
done
