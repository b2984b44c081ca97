#!/bin/bash
# fixed cost of a launch at short sequences: variants with the tile count of every workgroup capped (wrong results)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in n1k n4k c3; do timeout 200 python tools/variant_bench.py --config $c --rounds 5 --reps 10 main it0 ret1 ret2 ret2it0 2>&1 | grep median; done
