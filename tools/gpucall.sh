#!/bin/bash
# One entry point for what a gpurun call does on the GPU box (replaces the per-call scripts of round 2).  Runs from the repo root:
#     gpurun --timeout 900 -- 'bash tools/gpucall.sh <task> [args]'
# tasks
#   tests [pytest args]          python -m pytest tests -m gpu -q [args]          -> gpurun_out/<tag>/gpu_suite.log, parity_report.json
#   bench [bench.py args]        python bench.py [args]                            -> gpurun_out/<tag>/bench.json
#   prof  [bench.py args]        rocprofv3 --kernel-trace --stats of bench.py      -> gpurun_out/<tag>/kernel_stats.csv, bench_under_rocprof.json
#   pmc   <cfg>                  PMC passes over the attention kernel only (tools/run_kernel.py <cfg>; cfg may be c4 / c4nc)
#   pmcpp                        PMC passes over the one-launch pre-pass at the C3 shape (FETCH_SIZE / WRITE_SIZE / SQ)
#   ab    <cfg> tag1 tag2 ...    tools/variant_bench.py over libraries built by tools/build_variants.sh
# env: TAG (output directory under gpurun_out/, default r4)
set -u
task="${1:-}"; shift || true
tag="${TAG:-r4}"
out="gpurun_out/$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ATTN_CTRS=(
 "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VALU SQ_WAIT_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"
 "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM")
case "$task" in
  tests)
    timeout 850 python -m pytest tests -q -m gpu "$@" 2>&1 | grep -v "^  File\|^Extension\|Warning\|warnings.warn" | tail -15 | tee "$out/gpu_suite.log"
    cp gpurun_out/parity_report.json "$out/" 2>/dev/null ;;
  bench)
    python bench.py "$@" > "$out/bench.json" 2> "$out/bench.err"; tail -2 "$out/bench.err"; cut -c1-1500 "$out/bench.json" ;;
  prof)
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python bench.py "$@" > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err"
    f=$(ls "$out"/prof/*/*kernel_stats.csv | head -1); cp "$f" "$out/kernel_stats.csv"; head -8 "$out/kernel_stats.csv"; rm -rf "$out/prof"
    cut -c1-600 "$out/bench_under_rocprof.json" ;;
  pmc)
    cfg="${1:-c3}"
    SAGE_PMC_CFG=$cfg bash tools/pmc_passes.sh "$out/pmc_${cfg}" "${ATTN_CTRS[@]}" > "$out/pmc_${cfg}.txt" 2>&1
    cat "$out/pmc_${cfg}.txt" ;;
  pmcpp)
    bash tools/pmc_prepass.sh "$out/pmc_prepass" 2>&1 | tee "$out/pmc_prepass.txt" ;;
  ab)
    cfg="${1:-c3}"; shift
    timeout 500 python tools/variant_bench.py --config "$cfg" --rounds "${ROUNDS:-5}" --reps "${REPS:-5}" "$@" 2>&1 | tail -$(( 2 * $# + 2 )) | tee "$out/ab_$cfg.txt" ;;
  *) echo "usage: tools/gpucall.sh tests|bench|prof|pmc|pmcpp|ab ..."; exit 2 ;;
esac
