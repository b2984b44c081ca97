#!/bin/bash
# pre-pass A/B: main vs named variants, three shapes, twice
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for shape in 2,32,8192,128 2,48,17776,64 1,16,32768,128; do
    timeout 300 python tools/prepass_ab.py --shape $shape $TAGS 2>&1 | grep "median\|False"
done; done
