mkdir -p gpurun_out/r2m
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2m/prof -- python bench.py > gpurun_out/r2m/bench_under_rocprof.json 2> gpurun_out/r2m/bench_under_rocprof.err
f=$(ls gpurun_out/r2m/prof/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/r2m/c3_kernel_stats.csv; head -8 gpurun_out/r2m/c3_kernel_stats.csv; rm -rf gpurun_out/r2m/prof
cut -c1-600 gpurun_out/r2m/bench_under_rocprof.json
for cfg in c3 c5 c2; do
SAGE_PMC_CFG=$cfg bash tools/pmc_passes.sh gpurun_out/r2m/pmc_$cfg \
 "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VALU SQ_WAIT_INST_LDS" \
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
 "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
 > gpurun_out/r2m/pmc_$cfg.txt 2>&1
echo "== $cfg"; cat gpurun_out/r2m/pmc_$cfg.txt
done
