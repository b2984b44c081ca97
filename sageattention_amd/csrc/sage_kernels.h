// sage_kernels.h -- internal launch interface between the C ABI (sage_cabi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sage {

// query-scale granularity as seen by the attention kernel (slots per 128-row block)
enum : int { QG_PER_BLOCK = 1, QG_PER_WARP32 = 2, QG_PER_THREAD = 3, QG_PER_WARP16 = 4, QG_PER_THREAD16 = 5 };

constexpr int kAttnSchedBytes = 32 * 128;   // 32 ticket queues (8 XCDs x 4), one counter per 128-byte line
constexpr int kAttnSchedDoneWord = 16;      // (word 16 of line 0) workgroups that have left a persistent launch: the last one returns every word to zero

struct AttnParams {
    const void *q;            // int8 (or fp16 / bf16 for the fused-Q kernels), strides below (elements)
    const int8_t *k;
    const void *v;            // gfx950 tiled V^T image (see sage_prep_v.hip) -- or, v_rows != 0, the caller's fp16 V itself (rows of D halves)
    long v_sb, v_sh, v_sl;    // v_rows: element strides of that tensor (batch, kv-head, token)
    int v_rows;
    void *o;                  // fp16 / bf16
    float *lse;               // nullable, [B,Hq,Lq] log2 units
    const float *q_scale;
    const float *k_scale;
    const float *v_scale;     // [B,Hkv,D] (fp8 PV) or null
    const float *v_mean;      // [B,Hkv,D] or null
    const void *mask;         // attn_mask (bool bytes or fp16/bf16 additive), element strides below; null = none
    long m_sb, m_sh, m_sq, m_sk;
    const int32_t *cu_q;      // varlen only (null => dense)
    const int32_t *cu_k;
    const int32_t *cu_qs;     // prefix sums of ceil(Lq_i/128)
    const int32_t *cu_ks;     // prefix sums of ceil(Lk_i/64)
    const int32_t *seq_order; // varlen, nullable: permutation of sequence indices in processing order (legacy unit order, no work list)
    const int32_t *work_items;// varlen, nullable: the device-built work list of sage_varlen_plan ((sequence, query block), heaviest first)
    const int32_t *work_hdr;  // with work_items: {nitems, group, fold, left, ...} (kVarlenHdrWords)
    int items_bound;          // with work_items: host-known upper bound of nitems (sizes the grid)
    unsigned *sched;          // persistent launch (nullable): kAttnSchedBytes of ticket counters, ZERO on entry, zero again when the launch ends
    int nwg;                  // with sched: the logical grid (workgroup indices 0 .. nwg - 1 are dealt as tickets; the launch has fewer workgroups)
    int B, Hq, Hkv, group;    // group = Hq / Hkv
    int Lq, Lk;               // dense lengths; varlen: max lengths (grid sizing only)
    int nqblk;                // ceil(max Lq / 128)
    long q_sb, q_sh, q_sl;
    long k_sb, k_sh, k_sl;
    long o_sb, o_sh, o_sl;
    int nqs, nks;             // scale slots per (b,h) for q / k (dense)
    int qs_per_blk;           // q scale slots per 128-row block
    int q_gran;               // QG_*
    int kv_split;             // > 1: split-KV, the key range of a head is folded into the kv-head dimension (dense, fused-Q kernels)
    int ks_shift;             // k scale groups span 64 << ks_shift keys (1: the sm90 kernels' 128-key groups, core.py:964-970)
    int out_dtype;            // DT_F16 / DT_BF16
    long lse_sh;              // varlen lse head stride (unused for dense)
    float sm_scale_log2;      // multiplier taking dequantised scores to the log2 domain
    float q_premul;           // fused per-block Q quantisation (launch_attn_fused_qblock): q is multiplied by this before its abs-max
    int order_group;          // set by the launchers (sage_attn.hip set_work_order): causal dense work order, heads per group; 0 = head-major
    int order_fold;           // 1: single-round grid, pair the i-th longest with the i-th shortest block on a CU
    int order_left;           // (B * Hq) % 8 heads whose query blocks are dealt to all eight XCDs
    unsigned *trace;          // -DSAGE_ATTN_TRACE=1 builds (tools/attn_trace.py): 16 words per logical workgroup, nullable; ignored otherwise
    int trace_wgs;
};

// launch attributes of an attention call that are not kernel parameters (the C ABI's SageLaunchAttr, include/sage_gfx950.h)
struct AttnLaunchOpts {
    hipStream_t stream;
    bool fp8_folded;         // FP8 PV: the folded score form (opt-in variant) instead of the exact one
    bool force_persistent;   // take the ticket queues whatever the number of rounds (tests; AttnParams::sched must be set)
    int *grid_out;           // nullable (host): receives the number of workgroups launched
};
// mask_kind: 0 none, 1 bool, 2 additive fp16, 3 additive bf16 (FP16-PV, per-block scales, non-causal only)
hipError_t launch_attn(const AttnParams &p, int head_dim, bool pv_fp8, bool causal, bool kthread,
                       bool two_level, int mask_kind, const AttnLaunchOpts &o);
// q in fp16 / bf16, quantised per-thread in the kernel prologue; dense only.  FP8 PV: two-level accumulation; FP16 PV: FP32 accumulation
hipError_t launch_attn_fused_q(const AttnParams &p, int head_dim, bool causal, int q_dtype, bool pv_fp8, const AttnLaunchOpts &o);
// q in fp16 / bf16, quantised per 128-row block in the prologue after the multiplication by p.q_premul; FP16 PV in the Triton kernels'
// form, per-block k scales; dense or varlen (p.cu_q)
hipError_t launch_attn_fused_qblock(const AttnParams &p, int head_dim, bool causal, int q_dtype, const AttnLaunchOpts &o);

// causal dense work order of the 128-row kernels: -1 grouped / folded by grid size, 0 head-major, n groups of n heads (SAGE_ORDER_GROUP)
int work_order();
void set_work_order_mode(int group);

// ---- INT8 quantisation of Q / K ----------------------------------------------------------------
enum : int { QS_TRITON = 0, QS_CUDA = 1, QS_TRITON_THREAD = 2 };          // rounding / epsilon style
enum : int { GR_BLOCK = 1, GR_WARP = 2, GR_THREAD_Q = 3, GR_THREAD_K = 5 };  // row -> scale group map

struct QuantParams {
    const void *x;            // fp16 / bf16
    const void *mean;         // nullable [B,H,D] (strides mean_sb, mean_sh)
    int8_t *out;
    float *scale;
    const int32_t *cu;        // varlen: cu_seqlens (null => dense)
    const int32_t *cu_scale;  // varlen: prefix of blocks
    int B, H, L, D;
    long x_sb, x_sh, x_sl;
    long o_sb, o_sh, o_sl;
    long mean_sb, mean_sh;
    int nscale;               // dense: scale slots per (b,h)
    int blk;                  // rows per workgroup: 128 (Q) / 64 (K)
    int warp;                 // sub-group rows for GR_WARP / GR_THREAD_* (32, 16, 64)
    int gran, style, dtype;
    float pre_scale;
};
hipError_t launch_quant_int8(const QuantParams &p, hipStream_t stream);
// packed batches: every index array of a varlen call from one launch (sage_varlen_plan.hip), nseq <= kVarlenPlanMaxSeq
constexpr int kVarlenPlanMaxSeq = 1024;
constexpr int kVarlenHdrWords = 8;       // hdr: nitems, group, fold, left, nslab (K / V pre-pass slabs), max Lk, sum Lk, 0
struct VarlenPlanParams {
    const int32_t *cu_q, *cu_k;          // [nseq + 1]
    int nseq, blkq, blkk;
    int total_k;                         // rows of the packed k / v tensors (>= cu_k[nseq]): rows outside every sequence still count in the K mean
    int causal, Hq, Hkv, head_dim, pv_fp8;   // for the launch plan (hdr) only
    int32_t *cu_qs;                      // nullable [nseq + 1]
    int32_t *cu_ks;                      // [nseq + 1]
    int32_t *order;                      // nullable [nseq]
    int32_t *items;                      // nullable [2 * nitems]: (sequence, query block), heaviest first
    int32_t *slab_first;                 // nullable [nseq + 3]: prefix sums of the slab counts of the nseq sequences, then of two "gap" segments
                                         // (index nseq: rows cu_k[nseq] .. total_k, index nseq + 1: rows 0 .. cu_k[0]) that only the K mean reads
    int32_t *slab_seq;                   // nullable [nslab]: slab -> segment (sequence, or nseq / nseq + 1 for the gaps)
    int32_t *hdr;                        // nullable [kVarlenHdrWords]
    int forced_group;                    // work_order() when > 0 (experiments: heads per XCD group of the attention launch's plan), else -1
    int items_cap, slab_cap;             // (sequence, query block) pairs `items` holds, entries of `slab_seq`: counts derived from cu_seqlens on the
                                         // device are clamped to them (hdr reports the clamped counts), so an inconsistent cu_seqlens cannot write past
};
hipError_t launch_varlen_plan(const VarlenPlanParams &p, hipStream_t stream);

// ---- per-channel statistics over the sequence (K mean, V amax / mean) -----------------------------
#ifndef SAGE_STATS_SLAB
#define SAGE_STATS_SLAB 512
#endif
constexpr int kStatsSlab = SAGE_STATS_SLAB;   // tokens per stage-1 workgroup
struct StatsParams {
    const void *x;            // fp16 / bf16 [.., L, D] with strides
    float *ws;                // [B,H,nslab,3,D] partial (max, min, sum)
    float *stats;             // nullable [B,H,3,D] final (max, min, sum)
    void *mean_out;           // nullable [B,H,D] mean in the input dtype
    int B, H, L, D, nslab;
    long x_sb, x_sh, x_sl;
    int dtype;
    // packed batches (nullable, together): slabs per segment as sage_varlen_plan lays them out; nslab = host-known bound, L = rows of x
    const int32_t *cu, *slab_first, *slab_seq, *hdr;
    int nseq;
};
hipError_t launch_stats(const StatsParams &p, hipStream_t stream);

// ---- V pre-pass -------------------------------------------------------------------------------
struct PrepVParams {
    const void *v;            // fp16 / bf16 [.., L, D] with strides
    void *out;                // tiled V^T image, fp8 or fp16
    const float *stats;       // fp8: [B,H,3,D] (max, min, sum) from launch_stats
    const float *mean_in;     // fp16 path: nullable [B,H,D] mean to subtract (smooth_v)
    float *v_scale;           // [B,H,D] out (fp8)
    float *v_mean;            // fp8: nullable [B,H,D] out; non-null = smooth_v (subtract the mean)
    const int32_t *cu;        // varlen
    const int32_t *cu_tiles;  // varlen: prefix of ceil(L_i/64)
    int B, H, L, D;
    long v_sb, v_sh, v_sl;
    int dtype;
    int fp8;                  // 1: e4m3 output, 0: fp16 output
    float scale_max;
};
hipError_t launch_prep_v(const PrepVParams &p, hipStream_t stream);

// ---- fused K / V pre-pass: one launch, one read of K and V (sage_prepass.hip) -----------------------------------------
constexpr int kPrepassMaxSlabs = 65536 / kStatsSlab;   // slabs of one head that wait for each other inside the launch (L <= 65536):
                                                       // 16 per XCD against 64 resident workgroup slots per XCD
constexpr int kPrepassSyncStride = 32;  // uint32 words per (part, head): arrival + departure counter on their own 128-B line
struct PrepassParams {
    const void *k;            // fp16 / bf16 [.., L, D] with strides (parts & 1)
    const void *v;            // same shape (parts & 2)
    void *k_mean;             // nullable [B,H,D] out, input dtype; non-null = smooth_k (subtract it before quantising)
    int8_t *k_out;            // INT8 K, strides ko_*
    float *k_scale;           // [B,H,ceil(L/k_blk)*groups]
    void *v_image;            // tile image [B,H,ceil(L/64),D,64], fp8 or fp16 (v_fp16)
    float *v_scale;           // [B,H,D] out
    float *v_mean;            // nullable [B,H,D] out; non-null = smooth_v
    float *ws;                // [2,B,H,nslab,3,D] slab partials
    unsigned *sync;           // [2,B,H,kPrepassSyncStride] per head: arrival counter, departure counter, give-up flag.  ZERO on entry (the
                              // caller zeroes the buffer once); the kernel returns both counters to zero before it ends
    int B, H, L, D, nslab;
    long k_sb, k_sh, k_sl;
    long v_sb, v_sh, v_sl;
    long ko_sb, ko_sh, ko_sl;
    int parts;                // 1: K only, 2: V only, 3: both
    int v_fp16;               // V half: 0 = per-channel FP8 image (statistics + scales), 1 = fp16 image (`v.to(float16)`, no statistics)
    int k_blk;                // keys per scale block: 64 / 128
    int k_warp;               // == k_blk (one thread-group map per block)
    int k_gran;               // GR_BLOCK / GR_THREAD_K
    int k_style;              // QS_*
    int dtype;
    float scale_max;          // 448 for e4m3
    int debug_fail;           // test hook (sage_debug_prepass_fail): wait for one slab more than exists and give up after 2^10 polls
    unsigned *host_flag;      // nullable: device-visible pinned HOST word that a workgroup which gives up sets to 1 (system-scope store), so
                              // the caller can learn of a poisoned launch at its next call without a synchronisation
    // packed (varlen) batches: k / v are [sum L, H, D]; a slab is 512 tokens of ONE sequence, the head barrier and the K mean span all
    // sequences (core.py:432-434).  B = 1, L = sum L (the mean's divisor), nslab = host-known bound of the slab count (grid, workspace)
    const int32_t *cu;        // null => dense.  cu_seqlens_k [nseq + 1]
    const int32_t *cu_tiles;  // prefix sums of ceil(L_i / 64): k scale block / V tile index of a sequence's first block
    const int32_t *slab_seq;  // slab -> segment: a sequence, or a gap (nseq: rows cu[nseq] .. L, nseq + 1: rows 0 .. cu[0]; statistics only)
    const int32_t *slab_first;// prefix sums of the segments' slab counts
    const int32_t *hdr;       // sage_varlen_plan's header: hdr[4] = number of slabs that exist
    int nseq;                 // number of sequences (segments >= nseq are gaps)
};
hipError_t launch_prepass_kv(const PrepassParams &p, hipStream_t stream);
hipError_t launch_debug_spin(int ms, int nwg, hipStream_t stream);

// ---- LSE merge of partial attention states (sequence-parallel callers) -----------------------------
struct MergeParams {
    float *o_acc;             // [B,H,L,D] fp32 running output, contiguous
    float *lse_acc;           // [B,H,L] fp32 running log-sum-exp (natural log), contiguous
    const void *o_new;        // fp16 / bf16 partial output, element strides below
    const float *lse_new;     // [B,H,L] contiguous
    void *o_out;              // nullable: also write the merged output in the dtype of o_new (last step)
    int B, H, L, D;
    long n_sb, n_sh, n_sl;    // o_new strides
    long o_sb, o_sh, o_sl;    // o_out strides
    int dtype;
    int first;                // 1: initialise the running state from (o_new, lse_new)
    int cpr_pad;              // set by launch_merge_states: lanes per row (power of two >= D/8, <= 64)
};
hipError_t launch_merge_states(const MergeParams &p, hipStream_t stream);

// ---- split-KV merge (one attention call computed as S (+1) key-range chunks) -------------------------------------------
struct SplitMergeParams {
    const void *o_part;       // fp16 [B, Hkv, S, group, L, D] contiguous (H = Hkv * group)
    const float *lse_part;    // [B, Hkv, S, group, L] log2-domain log-sum-exp of each chunk (-inf: no visible key)
    const void *o_tail;       // nullable fp16 [B, H, L, D]: one more chunk (ragged tail of the key range)
    const float *lse_tail;    // [B, H, L]
    void *o_out;              // fp16 / bf16, element strides below
    float *lse_out;           // nullable [B, H, L] log2 domain
    int B, S, H, L, D, group;
    long o_sb, o_sh, o_sl;
    int dtype;                // of o_out
    int cpr_pad;              // set by the launcher
};
hipError_t launch_merge_split(const SplitMergeParams &p, hipStream_t stream);

}  // namespace sage
