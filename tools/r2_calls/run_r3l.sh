mkdir -p gpurun_out/r3l
for mode in 1 1 1 1 1 1; do
  SAGE_FUSE_Q16=$mode timeout 200 python -m pytest $(cat tools/r2_calls/upto_c2.txt | tr '\n' ' ') -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|AssertionError: V" | cut -c1-600 | sed "s/^/fuseq16=$mode: /"
done 2>&1 | tee gpurun_out/r3l/flaky_after_fix.txt
