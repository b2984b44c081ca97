// sage_cabi.hip -- extern "C" entry points declared in include/sage_gfx950.h.
// Validates arguments, fills the kernel parameter blocks, launches on the caller's stream.
#include "../../include/sage_gfx950.h"
#include "sage_common.h"
#include "sage_kernels.h"
#include "sage_work_order.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_launch(hipError_t e, const char *what)
{
    if (e != hipSuccess) return fail(SAGE_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SAGE_OK;
}

#define SAGE_REQUIRE(cond, ...) do { if (!(cond)) return fail(SAGE_EINVAL, __VA_ARGS__); } while (0)

struct MaskArg { const void *ptr; int kind; int64_t sb, sh, sq, sk; };

// SageLaunchAttr (nullable) -> the launch workspace and the launcher's options; the attributes are arguments of THIS call, nothing is kept
struct LaunchAttr { unsigned *ws; sage::AttnLaunchOpts opts; unsigned *trace; int trace_wgs; };
int read_attr(const SageLaunchAttr *attr, void *stream, bool takes_ws, LaunchAttr &out)
{
    out.ws = nullptr;
    out.opts = sage::AttnLaunchOpts{static_cast<hipStream_t>(stream), false, false, nullptr};
    out.trace = nullptr;
    out.trace_wgs = 0;
    if (attr == nullptr) return SAGE_OK;
    SageLaunchAttr a{};
    // struct_bytes is what the CALLER's struct holds: fewer bytes than ours (an older caller) are read as far as they go, more (a newer
    // caller) are ignored beyond what this library knows; 0 -- a caller that never set it -- is refused rather than guessed at
    SAGE_REQUIRE(attr->struct_bytes >= 8, "SageLaunchAttr.struct_bytes = %u: set it to sizeof(SageLaunchAttr) (at least the 8-byte header)", attr->struct_bytes);
    const size_t n = attr->struct_bytes > sizeof(SageLaunchAttr) ? sizeof(SageLaunchAttr) : attr->struct_bytes;
    memcpy(&a, attr, n);
    SAGE_REQUIRE((a.flags & ~(SAGE_ATTR_FP8_EXACT_SCORES | SAGE_ATTR_FP8_FOLDED_SCORES | SAGE_ATTR_FORCE_PERSISTENT)) == 0, "unknown SageLaunchAttr.flags 0x%x", a.flags);
    SAGE_REQUIRE((a.flags & (SAGE_ATTR_FP8_EXACT_SCORES | SAGE_ATTR_FP8_FOLDED_SCORES)) != (SAGE_ATTR_FP8_EXACT_SCORES | SAGE_ATTR_FP8_FOLDED_SCORES),
                 "SageLaunchAttr.flags asks for both FP8 score forms");
    SAGE_REQUIRE(a.launch_ws == nullptr || (a.launch_ws_bytes >= sage::kAttnSchedBytes && (reinterpret_cast<uintptr_t>(a.launch_ws) & 127u) == 0),
                 "the launch workspace is %d bytes, 128-byte aligned, zeroed (got %lld bytes at %p)", sage::kAttnSchedBytes,
                 (long long)a.launch_ws_bytes, a.launch_ws);
    SAGE_REQUIRE(!(a.flags & SAGE_ATTR_FORCE_PERSISTENT) || a.launch_ws != nullptr, "SAGE_ATTR_FORCE_PERSISTENT needs a launch workspace");
    out.ws = takes_ws ? static_cast<unsigned *>(a.launch_ws) : nullptr;
    out.opts.fp8_folded = (a.flags & SAGE_ATTR_FP8_FOLDED_SCORES) != 0;
    out.opts.force_persistent = takes_ws && (a.flags & SAGE_ATTR_FORCE_PERSISTENT) != 0;
    out.opts.grid_out = a.grid_out;
    out.trace = a.trace;
    out.trace_wgs = a.trace != nullptr ? a.trace_wgs : 0;
    return SAGE_OK;
}

int attn_common(bool fp8, bool varlen, const int8_t *q, const int8_t *k, const void *v_image, void *o, float *lse,
                const float *q_scale, const float *k_scale, const float *v_scale, const float *v_mean,
                const int32_t *cu_q, const int32_t *cu_k, const int32_t *cu_qs, const int32_t *cu_ks,
                int B, int Hq, int Hkv, int Lq, int Lk, int D,
                int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                int64_t o_sb, int64_t o_sh, int64_t o_sl,
                int is_causal, int gran, int q_warp, float sm_scale_log2, int pv_accum, int out_dtype, void *stream, const SageLaunchAttr *attr,
                const MaskArg *mask = nullptr, const int32_t *seq_order = nullptr,
                const int32_t *work_items = nullptr, const int32_t *work_hdr = nullptr, int items_bound = 0, const int64_t *v_strides = nullptr)
{
    LaunchAttr la;
    if (const int rc = read_attr(attr, stream, mask == nullptr, la)) return rc;
    if (v_strides != nullptr) {            // `v_image` is the caller's fp16 value tensor itself (rows), read in place
        SAGE_REQUIRE(!fp8 && !varlen && mask == nullptr, "V rows in place: dense, unmasked FP16-PV calls");
        SAGE_REQUIRE(v_strides[0] % 8 == 0 && v_strides[1] % 8 == 0 && v_strides[2] % 8 == 0 && v_strides[2] >= D, "v strides must be multiples of 8 elements (16-byte rows)");
        SAGE_REQUIRE(((int64_t)(Lk - 1) * v_strides[2] + D) * 2 < (int64_t)1 << 31, "one head of v must span less than 2 GiB");
    }
    SAGE_REQUIRE(q && k && v_image && o && q_scale && k_scale, "null tensor pointer");
    SAGE_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128 (got %d); pad on the host as core.py:260-271 does", D);
    SAGE_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Lq > 0, "empty problem (B=%d Hq=%d Hkv=%d Lq=%d)", B, Hq, Hkv, Lq);
    SAGE_REQUIRE(varlen || Lk > 0, "kv_len must be positive");
    SAGE_REQUIRE(Hq % Hkv == 0, "num_qo_heads (%d) must be divisible by num_kv_heads (%d)", Hq, Hkv);
    SAGE_REQUIRE(out_dtype == SAGE_DTYPE_F16 || out_dtype == SAGE_DTYPE_BF16, "bad out_dtype %d", out_dtype);
    const bool k128 = (gran & SAGE_GRAN_KBLK128) != 0;       // k scale groups of 128 keys (sm90 configuration)
    gran &= ~SAGE_GRAN_KBLK128;
    SAGE_REQUIRE(gran >= SAGE_GRAN_PER_BLOCK && gran <= SAGE_GRAN_PER_THREAD, "bad qk_quant_gran %d", gran);
    SAGE_REQUIRE(!k128 || (!varlen && mask == nullptr), "128-key k scale groups: dense, unmasked attention only");
    SAGE_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v_image) && aligned16(o), "q/k/v/o must be 16-byte aligned");
    SAGE_REQUIRE(q_sl % 16 == 0 && k_sl % 16 == 0 && q_sh % 16 == 0 && k_sh % 16 == 0 && q_sb % 16 == 0 && k_sb % 16 == 0,
                 "int8 q/k strides must be multiples of 16");
    SAGE_REQUIRE(o_sl % 8 == 0 && o_sh % 8 == 0 && o_sb % 8 == 0, "output strides must be multiples of 8 elements");
    SAGE_REQUIRE(!fp8 || v_scale, "fp8 PV needs v_scale");
    SAGE_REQUIRE(!varlen || (cu_q && cu_k && cu_qs && cu_ks), "varlen needs cu_seqlens arrays");

    sage::AttnParams p{};
    p.sched = la.ws; p.trace = la.trace; p.trace_wgs = la.trace_wgs;
    p.q = q; p.k = k; p.v = v_image; p.o = o; p.lse = lse;
    p.q_scale = q_scale; p.k_scale = k_scale; p.v_scale = v_scale; p.v_mean = v_mean;
    p.cu_q = cu_q; p.cu_k = cu_k; p.cu_qs = cu_qs; p.cu_ks = cu_ks; p.seq_order = varlen ? seq_order : nullptr;
    SAGE_REQUIRE((work_items == nullptr) == (work_hdr == nullptr) && (work_items == nullptr || (varlen && items_bound > 0)),
                 "the work list comes as (work_items, work_hdr, items_bound > 0), varlen only");
    p.work_items = work_items; p.work_hdr = work_hdr; p.items_bound = items_bound;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.group = Hq / Hkv;
    p.Lq = Lq; p.Lk = Lk;
    p.nqblk = (Lq + sage::BLKQ - 1) / sage::BLKQ;
    p.q_sb = q_sb; p.q_sh = q_sh; p.q_sl = q_sl;
    p.k_sb = k_sb; p.k_sh = k_sh; p.k_sl = k_sl;
    p.o_sb = o_sb; p.o_sh = o_sh; p.o_sl = o_sl;
    bool kthread = false;
    if (gran == SAGE_GRAN_PER_BLOCK) { p.q_gran = sage::QG_PER_BLOCK; p.qs_per_blk = 1; }
    else if (gran == SAGE_GRAN_PER_WARP) {
        SAGE_REQUIRE(q_warp == 32 || q_warp == 16, "per_warp q_warp must be 32 or 16 (got %d)", q_warp);
        p.q_gran = q_warp == 32 ? sage::QG_PER_WARP32 : sage::QG_PER_WARP16;
        p.qs_per_blk = sage::BLKQ / q_warp;
    } else {
        SAGE_REQUIRE(q_warp == 32 || q_warp == 16, "per_thread q_warp must be 32 or 16 (got %d)", q_warp);
        p.q_gran = q_warp == 32 ? sage::QG_PER_THREAD : sage::QG_PER_THREAD16;
        p.qs_per_blk = (sage::BLKQ / q_warp) * 8; kthread = true;
    }
    SAGE_REQUIRE(!varlen || gran == SAGE_GRAN_PER_BLOCK, "varlen supports per_block scales only");
    p.nqs = p.nqblk * p.qs_per_blk;
    p.ks_shift = k128 ? 1 : 0;
    p.nks = ((Lk + (sage::BLKK << p.ks_shift) - 1) / (sage::BLKK << p.ks_shift)) * (kthread ? 4 : 1);
    p.out_dtype = out_dtype;
    p.lse_sh = 0;
    p.sm_scale_log2 = sm_scale_log2;
    if (v_strides != nullptr) { p.v_rows = 1; p.v_sb = v_strides[0]; p.v_sh = v_strides[1]; p.v_sl = v_strides[2]; }
    int mask_kind = 0;
    if (mask != nullptr) {
        SAGE_REQUIRE(mask->ptr, "null attn_mask pointer");
        SAGE_REQUIRE(mask->kind >= SAGE_MASK_BOOL && mask->kind <= SAGE_MASK_BF16, "bad mask_kind %d", mask->kind);
        SAGE_REQUIRE(!is_causal, "Mask should be None for causal attention.");           // core.py:310
        p.mask = mask->ptr; p.m_sb = mask->sb; p.m_sh = mask->sh; p.m_sq = mask->sq; p.m_sk = mask->sk;
        mask_kind = mask->kind;
    }
    SAGE_REQUIRE(pv_accum >= SAGE_PV_ACCUM_SINGLE && pv_accum <= SAGE_PV_ACCUM_TRITON && (!fp8 || pv_accum != SAGE_PV_ACCUM_TRITON),
                 "bad pv_accum %d", pv_accum);
    // FP16 PV: the kernel's TWO_LEVEL parameter selects the Triton kernel form (true) or the CUDA kernel form (false)
    const bool two_level = fp8 ? pv_accum == SAGE_PV_ACCUM_TWO_LEVEL : pv_accum == SAGE_PV_ACCUM_TRITON;
    return check_launch(sage::launch_attn(p, D, fp8, is_causal != 0, kthread, two_level, mask_kind, la.opts), "sage_attn launch");
}

}  // namespace

extern "C" {

SAGE_API int sage_abi_version(void) { return SAGE_ABI_VERSION; }
SAGE_API const char *sage_last_error(void) { return g_err; }
// host-side views of sage_work_order.h (the code the kernels and launchers run), for tests without a GPU
SAGE_API int sage_debug_work_order_plan(int nheads, int nqblk, int64_t kv_len, int head_dim, int pv_fp8, int forced, int *group, int *fold, int *left)
{
    if (nheads <= 0 || nqblk <= 0 || group == nullptr || fold == nullptr || left == nullptr) return fail(SAGE_EINVAL, "sage_debug_work_order_plan: bad argument");
    sage::WorkOrder w;
    const int grid = sage::plan_work_order(w, nheads, nqblk, (long)kv_len, head_dim, pv_fp8 != 0, forced);
    *group = w.group; *fold = w.fold; *left = w.left;
    return grid;
}
SAGE_API int sage_debug_work_item(int bid, int nwg, int nheads, int nqblk, int group, int fold, int left, int *head, int *qrank)
{
    if (head == nullptr || qrank == nullptr || nwg <= 0 || bid < 0 || bid >= nwg || nheads <= 0 || nqblk <= 0 || group < 0 || fold < 0 || fold > 1 ||
        left < 0 || left >= 8 || left > nheads)
        return fail(SAGE_EINVAL, "sage_debug_work_item: bad argument");
    const sage::WorkOrder w = {group, fold, left};
    return sage::work_item(w, bid, nwg, nheads, nqblk, *head, *qrank) ? 1 : 0;
}
SAGE_API int sage_work_order(void) { return sage::work_order(); }
SAGE_API void sage_set_work_order(int group) { sage::set_work_order_mode(group < -1 ? -1 : group); }

SAGE_API int64_t sage_v_image_bytes(int head_dim, int fp8, int64_t n_kv_tiles_total)
{
    return n_kv_tiles_total * (int64_t)head_dim * (fp8 ? 64 : 128);
}

SAGE_API int sage_quant_qk_int8(const void *x, const void *mean, int8_t *out, float *scale,
                       int B, int H, int L, int D,
                       int64_t x_sb, int64_t x_sh, int64_t x_sl,
                       int64_t o_sb, int64_t o_sh, int64_t o_sl,
                       int64_t mean_sb, int64_t mean_sh,
                       int blk, int warp, int gran, int is_key, int style,
                       float pre_scale, int dtype, void *stream)
{
    SAGE_REQUIRE(x && out && scale, "null tensor pointer");
    SAGE_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128 (got %d)", D);
    SAGE_REQUIRE(blk == 64 || blk == 128, "blk must be 64 or 128 (got %d)", blk);
    SAGE_REQUIRE(B > 0 && H > 0 && L > 0, "empty tensor");
    SAGE_REQUIRE(dtype == SAGE_DTYPE_F16 || dtype == SAGE_DTYPE_BF16, "bad dtype %d", dtype);
    SAGE_REQUIRE(style >= 0 && style <= 2, "bad style %d", style);
    SAGE_REQUIRE(aligned16(x) && aligned16(out) && (!mean || aligned16(mean)), "x/out/mean must be 16-byte aligned");
    SAGE_REQUIRE(x_sl % 8 == 0 && x_sh % 8 == 0 && x_sb % 8 == 0, "input strides must be multiples of 8 elements");
    SAGE_REQUIRE(o_sl % 16 == 0 && o_sh % 16 == 0 && o_sb % 16 == 0, "int8 output strides must be multiples of 16");
    SAGE_REQUIRE(!mean || (mean_sb % 8 == 0 && mean_sh % 8 == 0), "mean strides must be multiples of 8 elements");
    sage::QuantParams p{};
    p.x = x; p.mean = mean; p.out = out; p.scale = scale; p.cu = nullptr; p.cu_scale = nullptr;
    p.B = B; p.H = H; p.L = L; p.D = D;
    p.x_sb = x_sb; p.x_sh = x_sh; p.x_sl = x_sl; p.o_sb = o_sb; p.o_sh = o_sh; p.o_sl = o_sl;
    p.mean_sb = mean_sb; p.mean_sh = mean_sh;
    p.blk = blk; p.warp = warp; p.style = style; p.dtype = dtype; p.pre_scale = pre_scale;
    int slots = 1;
    if (gran == SAGE_GRAN_PER_BLOCK) { p.gran = sage::GR_BLOCK; p.warp = blk; }
    else if (gran == SAGE_GRAN_PER_WARP) {
        SAGE_REQUIRE(warp > 0 && blk % warp == 0 && blk / warp <= 32, "bad warp block %d for blk %d", warp, blk);
        p.gran = sage::GR_WARP; slots = blk / warp;
    } else if (gran == SAGE_GRAN_PER_THREAD) {
        SAGE_REQUIRE(warp > 0 && blk % warp == 0 && warp % 8 == 0, "bad warp block %d for blk %d", warp, blk);
        p.gran = is_key ? sage::GR_THREAD_K : sage::GR_THREAD_Q;
        slots = (blk / warp) * (is_key ? 4 : 8);
        SAGE_REQUIRE(slots <= 64, "too many scale groups per block (%d)", slots);
    } else return fail(SAGE_EINVAL, "bad qk_quant_gran %d", gran);
    p.nscale = ((L + blk - 1) / blk) * slots;
    return check_launch(sage::launch_quant_int8(p, static_cast<hipStream_t>(stream)), "sage_quant_qk_int8 launch");
}

SAGE_API int sage_quant_qk_int8_varlen(const void *x, const void *mean, int8_t *out, float *scale,
                              const int32_t *cu_seqlens, const int32_t *cu_scale,
                              int nseq, int max_seqlen, int H, int D,
                              int64_t x_sl, int64_t x_sh, int64_t o_sl, int64_t o_sh, int64_t mean_sh,
                              int blk, float pre_scale, int dtype, void *stream)
{
    SAGE_REQUIRE(x && out && scale && cu_seqlens && cu_scale, "null tensor pointer");
    SAGE_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128 (got %d)", D);
    SAGE_REQUIRE(blk == 64 || blk == 128, "blk must be 64 or 128 (got %d)", blk);
    SAGE_REQUIRE(nseq > 0 && H > 0 && max_seqlen > 0, "empty batch");
    SAGE_REQUIRE(dtype == SAGE_DTYPE_F16 || dtype == SAGE_DTYPE_BF16, "bad dtype %d", dtype);
    SAGE_REQUIRE(aligned16(x) && aligned16(out) && (!mean || aligned16(mean)), "x/out/mean must be 16-byte aligned");
    SAGE_REQUIRE(x_sl % 8 == 0 && x_sh % 8 == 0 && o_sl % 16 == 0 && o_sh % 16 == 0, "bad strides");
    sage::QuantParams p{};
    p.x = x; p.mean = mean; p.out = out; p.scale = scale; p.cu = cu_seqlens; p.cu_scale = cu_scale;
    p.B = nseq; p.H = H; p.L = max_seqlen; p.D = D;
    p.x_sb = 0; p.x_sh = x_sh; p.x_sl = x_sl; p.o_sb = 0; p.o_sh = o_sh; p.o_sl = o_sl;
    p.mean_sb = 0; p.mean_sh = mean_sh;
    p.blk = blk; p.warp = blk; p.gran = sage::GR_BLOCK; p.style = sage::QS_TRITON; p.dtype = dtype;
    p.pre_scale = pre_scale; p.nscale = 0;
    return check_launch(sage::launch_quant_int8(p, static_cast<hipStream_t>(stream)), "sage_quant_qk_int8_varlen launch");
}

SAGE_API int sage_varlen_plan_max_seqs(void) { return sage::kVarlenPlanMaxSeq; }
SAGE_API int sage_varlen_plan(const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k, int nseq, int total_k, int blkq, int blkk,
                              int is_causal, int Hq, int Hkv, int head_dim, int pv_fp8,
                              int32_t *cu_q_scale, int32_t *cu_k_scale, int32_t *seq_order,
                              int32_t *work_items, int work_items_cap, int32_t *slab_first, int32_t *slab_seq, int slab_seq_cap,
                              int32_t *hdr, void *stream)
{
    SAGE_REQUIRE((work_items == nullptr || work_items_cap > 0) && (slab_seq == nullptr || slab_seq_cap > 0),
                 "work_items / slab_seq come with their capacities (got %d, %d)", work_items_cap, slab_seq_cap);
    SAGE_REQUIRE(cu_seqlens_q && cu_seqlens_k && cu_k_scale, "null tensor pointer");
    SAGE_REQUIRE(nseq > 0 && nseq <= sage::kVarlenPlanMaxSeq, "nseq must be in 1 .. %d (got %d)", sage::kVarlenPlanMaxSeq, nseq);
    SAGE_REQUIRE(blkq > 0 && blkk > 0, "block sizes must be positive");
    SAGE_REQUIRE(work_items == nullptr || (hdr != nullptr && blkq == sage::BLKQ && blkk == sage::BLKK),
                 "the work list needs hdr and the attention kernel's blocks (%d query rows, %d keys)", sage::BLKQ, sage::BLKK);
    SAGE_REQUIRE(hdr == nullptr || (Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && (head_dim == 64 || head_dim == 128)),
                 "the launch plan needs Hq %% Hkv == 0 and head_dim 64 / 128 (got %d, %d, %d)", Hq, Hkv, head_dim);
    SAGE_REQUIRE((slab_seq == nullptr) == (slab_first == nullptr) && (slab_seq == nullptr || hdr != nullptr), "slab_seq, slab_first and hdr come together");
    sage::VarlenPlanParams p{};
    SAGE_REQUIRE(slab_seq == nullptr || total_k > 0, "the slab map needs total_k, the row count of the packed k / v tensors");
    p.cu_q = cu_seqlens_q; p.cu_k = cu_seqlens_k; p.nseq = nseq; p.blkq = blkq; p.blkk = blkk; p.total_k = total_k;
    p.causal = is_causal ? 1 : 0; p.Hq = Hq > 0 ? Hq : 1; p.Hkv = Hkv > 0 ? Hkv : 1; p.head_dim = head_dim; p.pv_fp8 = pv_fp8 ? 1 : 0;
    p.cu_qs = cu_q_scale; p.cu_ks = cu_k_scale; p.order = seq_order; p.items = work_items;
    p.slab_first = slab_first; p.slab_seq = slab_seq; p.hdr = hdr;
    p.items_cap = work_items_cap; p.slab_cap = slab_seq_cap;
    p.forced_group = sage::work_order() > 0 ? sage::work_order() : -1;
    return check_launch(sage::launch_varlen_plan(p, static_cast<hipStream_t>(stream)), "sage_varlen_plan launch");
}
// host-side view of the work list (csrc/sage_work_order.h, the functions varlen_plan_kernel runs; no GPU needed): lq / lk are HOST arrays of
// the nseq sequence lengths; items_out receives 2 * nitems ints ((sequence, query block), heaviest first), hdr_out {nitems, group, fold, left};
// returns the grid size of the launch, or a negative status
SAGE_API int sage_debug_varlen_items(const int32_t *lq, const int32_t *lk, int nseq, int is_causal, int Hq, int Hkv, int head_dim, int pv_fp8,
                                     int32_t *items_out, int items_cap, int32_t *hdr_out)
{
    if (lq == nullptr || lk == nullptr || items_out == nullptr || hdr_out == nullptr || nseq <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 ||
        (head_dim != 64 && head_dim != 128))
        return fail(SAGE_EINVAL, "sage_debug_varlen_items: bad argument");
    int nitems = 0, max_lk = 0;
    for (int s = 0; s < nseq; s++) {
        if (lq[s] < 0 || lk[s] < 0) return fail(SAGE_EINVAL, "sage_debug_varlen_items: negative length");
        nitems += (lq[s] + 127) / 128;
        max_lk = lk[s] > max_lk ? lk[s] : max_lk;
    }
    if (nitems > items_cap) return fail(SAGE_EINVAL, "sage_debug_varlen_items: %d items do not fit %d", nitems, items_cap);
    for (int s = 0; s < nseq; s++)
        for (int j = 0; j < (lq[s] + 127) / 128; j++) {
            const int r = sage::varlen_item_rank(lq, lk, nseq, s, j, is_causal != 0);
            if (r < 0 || r >= nitems) return fail(SAGE_ELAUNCH, "sage_debug_varlen_items: rank %d out of range", r);
            items_out[2 * r] = s; items_out[2 * r + 1] = j;
        }
    sage::WorkOrder w;
    const int grid = sage::plan_varlen_order(w, Hq, Hq / Hkv, nitems, (long)max_lk, head_dim, pv_fp8 != 0);
    hdr_out[0] = nitems; hdr_out[1] = w.group; hdr_out[2] = w.fold; hdr_out[3] = w.left;
    return grid;
}

static int stats_common(const void *x, void *mean_out, float *ws, float *stats, int B, int H, int L, int D,
                        int64_t x_sb, int64_t x_sh, int64_t x_sl, int dtype, void *stream, const char *what)
{
    SAGE_REQUIRE(x && ws, "null tensor pointer");
    SAGE_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128 (got %d)", D);
    SAGE_REQUIRE(B > 0 && H > 0 && L > 0, "empty tensor");
    SAGE_REQUIRE(dtype == SAGE_DTYPE_F16 || dtype == SAGE_DTYPE_BF16, "bad dtype %d", dtype);
    SAGE_REQUIRE(aligned16(x), "input must be 16-byte aligned");
    SAGE_REQUIRE(x_sl % 8 == 0 && x_sh % 8 == 0 && x_sb % 8 == 0, "strides must be multiples of 8 elements");
    sage::StatsParams p{};
    p.x = x; p.ws = ws; p.stats = stats; p.mean_out = mean_out;
    p.B = B; p.H = H; p.L = L; p.D = D; p.nslab = (L + sage::kStatsSlab - 1) / sage::kStatsSlab;
    p.x_sb = x_sb; p.x_sh = x_sh; p.x_sl = x_sl; p.dtype = dtype;
    return check_launch(sage::launch_stats(p, static_cast<hipStream_t>(stream)), what);
}

static int prep_v_common(const void *v, void *v_image, float *v_scale, float *v_mean_out, const float *v_mean_in,
                         const float *stats, const int32_t *cu, const int32_t *cu_tiles, int B, int H, int L, int D,
                         int64_t v_sb, int64_t v_sh, int64_t v_sl, float scale_max, int dtype, int fp8, void *stream)
{
    SAGE_REQUIRE(v && v_image, "null tensor pointer");
    SAGE_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128 (got %d)", D);
    SAGE_REQUIRE(B > 0 && H > 0 && L > 0, "empty tensor");
    SAGE_REQUIRE(dtype == SAGE_DTYPE_F16 || dtype == SAGE_DTYPE_BF16, "bad dtype %d", dtype);
    SAGE_REQUIRE(aligned16(v) && aligned16(v_image), "v / v_image must be 16-byte aligned");
    SAGE_REQUIRE(v_sl % 8 == 0 && v_sh % 8 == 0 && v_sb % 8 == 0, "v strides must be multiples of 8 elements");
    sage::PrepVParams p{};
    p.v = v; p.out = v_image; p.stats = stats; p.mean_in = v_mean_in; p.v_scale = v_scale; p.v_mean = v_mean_out;
    p.cu = cu; p.cu_tiles = cu_tiles;
    p.B = B; p.H = H; p.L = L; p.D = D; p.v_sb = v_sb; p.v_sh = v_sh; p.v_sl = v_sl;
    p.dtype = dtype; p.fp8 = fp8; p.scale_max = scale_max;
    return check_launch(sage::launch_prep_v(p, static_cast<hipStream_t>(stream)), "sage_prep_v launch");
}

SAGE_API int64_t sage_stats_ws_floats(int B, int H, int L, int D)
{
    const int64_t nslab = (L + sage::kStatsSlab - 1) / sage::kStatsSlab;
    return (int64_t)B * H * (nslab + 1) * 3 * D;
}

SAGE_API int sage_channel_mean(const void *x, void *mean_out, float *ws, int B, int H, int L, int D,
                               int64_t x_sb, int64_t x_sh, int64_t x_sl, int dtype, void *stream)
{
    SAGE_REQUIRE(mean_out, "null output pointer");
    return stats_common(x, mean_out, ws, nullptr, B, H, L, D, x_sb, x_sh, x_sl, dtype, stream, "sage_channel_mean launch");
}

SAGE_API int sage_channel_mean_varlen(const void *x, void *mean_out, float *ws, const int32_t *cu_seqlens, const int32_t *slab_first,
                                      const int32_t *slab_seq, const int32_t *hdr, int nseq, int total_tokens, int nslab_bound, int H, int D,
                                      int64_t x_sl, int64_t x_sh, int dtype, void *stream)
{
    SAGE_REQUIRE(x && ws && mean_out && cu_seqlens && slab_first && slab_seq && hdr && nseq > 0, "null tensor pointer");
    SAGE_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128 (got %d)", D);
    SAGE_REQUIRE(H > 0 && total_tokens > 0 && nslab_bound >= (total_tokens + sage::kStatsSlab - 1) / sage::kStatsSlab, "empty tensor or nslab_bound too small");
    SAGE_REQUIRE(dtype == SAGE_DTYPE_F16 || dtype == SAGE_DTYPE_BF16, "bad dtype %d", dtype);
    SAGE_REQUIRE(aligned16(x) && x_sl % 8 == 0 && x_sh % 8 == 0, "input must be 16-byte aligned with strides in multiples of 8 elements");
    sage::StatsParams p{};
    p.x = x; p.ws = ws; p.stats = nullptr; p.mean_out = mean_out;
    p.B = 1; p.H = H; p.L = total_tokens; p.D = D; p.nslab = nslab_bound;
    p.x_sb = 0; p.x_sh = x_sh; p.x_sl = x_sl; p.dtype = dtype;
    p.cu = cu_seqlens; p.slab_first = slab_first; p.slab_seq = slab_seq; p.hdr = hdr; p.nseq = nseq;
    return check_launch(sage::launch_stats(p, static_cast<hipStream_t>(stream)), "sage_channel_mean_varlen launch");
}

SAGE_API int sage_prep_v_fp8(const void *v, void *v_image, float *v_scale, float *v_mean, float *ws,
                    int B, int H, int L, int D, int64_t v_sb, int64_t v_sh, int64_t v_sl,
                    float scale_max, int dtype, void *stream)
{
    SAGE_REQUIRE(v_scale && ws, "fp8 V pre-pass needs v_scale and the statistics workspace");
    SAGE_REQUIRE(scale_max > 0.0f, "scale_max must be positive");
    const int64_t nslab = (L + sage::kStatsSlab - 1) / sage::kStatsSlab;
    float *stats = ws + (int64_t)B * H * nslab * 3 * D;      // final block lives behind the partials
    int rc = stats_common(v, nullptr, ws, stats, B, H, L, D, v_sb, v_sh, v_sl, dtype, stream, "sage_v_stats launch");
    if (rc != SAGE_OK) return rc;
    return prep_v_common(v, v_image, v_scale, v_mean, nullptr, stats, nullptr, nullptr, B, H, L, D, v_sb, v_sh, v_sl,
                         scale_max, dtype, 1, stream);
}

SAGE_API int64_t sage_prepass_ws_floats(int B, int H, int L, int D)
{
    const int64_t nslab = (L + sage::kStatsSlab - 1) / sage::kStatsSlab;
    return 2 * (int64_t)B * H * nslab * 3 * D;
}

SAGE_API int64_t sage_prepass_sync_words(int B, int H) { return 2 * (int64_t)B * H * sage::kPrepassSyncStride; }

// Longest head the in-launch barrier takes on the current device: at most kPrepassMaxSlabs slabs, and never more than one per
// compute unit of the device (or partition) the caller runs on -- every slab of a head must be able to be resident while its
// head-mates arrive, with room left for the workgroups of the heads before it (2 resident workgroups per CU at D = 128).
static int g_prepass_debug_fail = 0;
SAGE_API void sage_debug_prepass_fail(int on) { g_prepass_debug_fail = on ? 1 : 0; }

// compute units a launch on `stream` can use: the stream's CU mask if it has one (hipExtStreamCreateWithCUMask), else the device's
static int stream_cu_count(hipStream_t stream, int dev)
{
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    uint32_t mask[16] = {0};
    if (hipExtStreamGetCUMask(stream, 16, mask) == hipSuccess) {
        int bits = 0;
        for (int i = 0; i < 16; i++) bits += __builtin_popcount(mask[i]);
        if (bits > 0 && bits < cus) cus = bits;
    }
    (void)hipGetLastError();
    return cus;
}

SAGE_API int sage_prepass_max_seqlen(void)
{
    static thread_local int cached_dev = -1, cached_len = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        const int slabs = cus < sage::kPrepassMaxSlabs ? cus : sage::kPrepassMaxSlabs;
        cached_len = slabs * sage::kStatsSlab;
        cached_dev = dev;
    }
    return cached_len;
}

// one device-visible word of pinned host memory for the `host_flag` argument below (owned by the caller: the library keeps no handle)
SAGE_API int sage_host_word_alloc(void **host_ptr, void **device_ptr)
{
    SAGE_REQUIRE(host_ptr && device_ptr, "null output pointer");
    void *h = nullptr, *d = nullptr;
    hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable);
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) { if (h) (void)hipHostFree(h); return fail(SAGE_ELAUNCH, "sage_host_word_alloc: %s", hipGetErrorString(e)); }
    *static_cast<volatile uint32_t *>(h) = 0u;
    *host_ptr = h; *device_ptr = d;
    return SAGE_OK;
}
SAGE_API int sage_host_word_free(void *host_ptr)
{
    if (host_ptr == nullptr) return SAGE_OK;
    const hipError_t e = hipHostFree(host_ptr);
    return e == hipSuccess ? SAGE_OK : fail(SAGE_ELAUNCH, "sage_host_word_free: %s", hipGetErrorString(e));
}

SAGE_API int sage_prepass_max_seqlen_stream(void *stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const int cus = stream_cu_count(static_cast<hipStream_t>(stream), dev);
    const int slabs = cus < sage::kPrepassMaxSlabs ? cus : sage::kPrepassMaxSlabs;
    return slabs * sage::kStatsSlab;
}

SAGE_API int sage_prepass_failed_heads(const uint32_t *sync, int B, int H, void *stream)
{
    if (!sync || B <= 0 || H <= 0) return fail(SAGE_EINVAL, "bad arguments");
    const size_t words = (size_t)2 * B * H * sage::kPrepassSyncStride;
    uint32_t *host = static_cast<uint32_t *>(malloc(words * sizeof(uint32_t)));
    if (!host) return fail(SAGE_ELAUNCH, "out of host memory");
    hipError_t e = hipMemcpyAsync(host, sync, words * sizeof(uint32_t), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    int n = 0;
    if (e == hipSuccess)
        for (size_t i = 0; i < (size_t)2 * B * H; i++) n += host[i * sage::kPrepassSyncStride + 2] != 0;
    free(host);
    if (e != hipSuccess) return fail(SAGE_ELAUNCH, "sage_prepass_failed_heads: %s", hipGetErrorString(e));
    return n;
}

SAGE_API int sage_prepass_kv(const void *k, const void *v, void *k_mean, int8_t *k_int8, float *k_scale,
                    void *v_image, float *v_scale, float *v_mean, float *ws, uint32_t *sync,
                    int B, int H, int L, int D,
                    int64_t k_sb, int64_t k_sh, int64_t k_sl, int64_t v_sb, int64_t v_sh, int64_t v_sl,
                    int64_t ko_sb, int64_t ko_sh, int64_t ko_sl,
                    int k_blk, int qk_quant_gran, int k_style, float scale_max, int v_fp16, int dtype, uint32_t *host_flag, void *stream)
{
    SAGE_REQUIRE(k || v, "nothing to do: both k and v are null");
    SAGE_REQUIRE(ws && sync, "the fused pre-pass needs its workspace and its sync buffer");
    SAGE_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128 (got %d)", D);
    SAGE_REQUIRE(B > 0 && H > 0 && L > 0, "empty tensor");
    SAGE_REQUIRE(B <= 32767 && H <= 65535, "batch / head count too large for one launch (%d, %d)", B, H);
    SAGE_REQUIRE(L <= sage_prepass_max_seqlen(), "sequence too long for the in-launch head barrier (%d > %d): use the "
                 "sage_channel_mean / sage_quant_qk_int8 / sage_prep_v_fp8 sequence", L, sage_prepass_max_seqlen());
    {   // the slabs of a head must be co-resident: a stream restricted to a CU mask has fewer compute units than the device
        int dev = 0;
        SAGE_REQUIRE(hipGetDevice(&dev) == hipSuccess, "no current device");
        const int cus = stream_cu_count(static_cast<hipStream_t>(stream), dev);
        const int nslab_ = (L + sage::kStatsSlab - 1) / sage::kStatsSlab;
        SAGE_REQUIRE(nslab_ <= cus, "a head of %d slabs cannot be co-resident on the %d compute units this stream may use: use the "
                     "sage_channel_mean / sage_quant_qk_int8 / sage_prep_v_fp8 sequence", nslab_, cus);
    }
    SAGE_REQUIRE(dtype == SAGE_DTYPE_F16 || dtype == SAGE_DTYPE_BF16, "bad dtype %d", dtype);
    // rows of the last slab past L are read through the buffer range check: their 32-bit byte offsets must not wrap either
    const int64_t lpad = ((int64_t)L + sage::kStatsSlab - 1) / sage::kStatsSlab * sage::kStatsSlab;
    sage::PrepassParams p{};
    p.parts = (k ? 1 : 0) | (v ? 2 : 0);
    if (k) {
        SAGE_REQUIRE(k_int8 && k_scale, "K part needs k_int8 and k_scale");
        SAGE_REQUIRE(aligned16(k) && aligned16(k_int8), "k / k_int8 must be 16-byte aligned");
        SAGE_REQUIRE(k_sl % 8 == 0 && k_sh % 8 == 0 && k_sb % 8 == 0, "input strides must be multiples of 8 elements");
        SAGE_REQUIRE(ko_sl % 16 == 0 && ko_sh % 16 == 0 && ko_sb % 16 == 0, "int8 output strides must be multiples of 16");
        SAGE_REQUIRE(((lpad - 1) * k_sl + D) * 2 < (int64_t)1 << 32 && (lpad - 1) * ko_sl + D < (int64_t)1 << 32,
                     "one head of k (rounded up to whole 512-row slabs) spans 4 GiB or more: the kernel addresses a head with 32-bit buffer offsets");
        SAGE_REQUIRE(k_blk == 64 || k_blk == 128, "k_blk must be 64 or 128 (got %d)", k_blk);
        SAGE_REQUIRE(k_style == sage::QS_CUDA || k_style == sage::QS_TRITON_THREAD || k_style == sage::QS_TRITON,
                     "k_style must be the Triton per-block (0), the CUDA (1) or the per-thread Triton (2) convention (got %d)", k_style);
        SAGE_REQUIRE(k_style != sage::QS_TRITON || qk_quant_gran == SAGE_GRAN_PER_BLOCK, "the Triton per-block convention goes with per-block scales");
        if (qk_quant_gran == SAGE_GRAN_PER_BLOCK) p.k_gran = sage::GR_BLOCK;
        else if (qk_quant_gran == SAGE_GRAN_PER_THREAD) p.k_gran = sage::GR_THREAD_K;
        else return fail(SAGE_EINVAL, "bad k granularity %d (per-block or per-thread)", qk_quant_gran);
    }
    if (v) {
        SAGE_REQUIRE(v_image && (v_scale || v_fp16), "V part needs v_image (and v_scale for the FP8 image)");
        SAGE_REQUIRE(aligned16(v) && aligned16(v_image), "v / v_image must be 16-byte aligned");
        SAGE_REQUIRE(v_sl % 8 == 0 && v_sh % 8 == 0 && v_sb % 8 == 0, "input strides must be multiples of 8 elements");
        SAGE_REQUIRE(((lpad - 1) * v_sl + D) * 2 < (int64_t)1 << 32,
                     "one head of v (rounded up to whole 512-row slabs) spans 4 GiB or more: the kernel addresses a head with 32-bit buffer offsets");
        SAGE_REQUIRE(v_fp16 || scale_max > 0.0f, "scale_max must be positive");
        SAGE_REQUIRE(!(v_fp16 && v_mean), "the fp16 image has no smooth_v (use sage_prep_v_f16 with a mean for sub_mean)");
    }
    p.k = k; p.v = v; p.k_mean = k_mean; p.k_out = k_int8; p.k_scale = k_scale;
    p.v_image = v_image; p.v_scale = v_scale; p.v_mean = v_mean; p.ws = ws; p.sync = sync;
    p.B = B; p.H = H; p.L = L; p.D = D; p.nslab = (L + sage::kStatsSlab - 1) / sage::kStatsSlab;
    p.k_sb = k_sb; p.k_sh = k_sh; p.k_sl = k_sl; p.v_sb = v_sb; p.v_sh = v_sh; p.v_sl = v_sl;
    p.ko_sb = ko_sb; p.ko_sh = ko_sh; p.ko_sl = ko_sl;
    p.k_blk = k_blk; p.k_warp = k_blk; p.k_style = k_style; p.dtype = dtype; p.scale_max = scale_max; p.v_fp16 = v_fp16 ? 1 : 0;
    p.debug_fail = g_prepass_debug_fail;
    p.host_flag = host_flag;
    // the per-head counters and give-up flags start from zero in every launch (a launch that gave up leaves them dirty); launch_prepass_kv
    // zeroes them with a small kernel of its own (a hipMemsetAsync node replayed wrongly inside a captured HIP graph on ROCm 7.0)
    return check_launch(sage::launch_prepass_kv(p, static_cast<hipStream_t>(stream)), "sage_prepass_kv launch");
}

// packed (varlen) batches: K mean over all packed tokens + per-sequence INT8 K (Triton per-block rounding) + fp16 V tile image, one launch
SAGE_API int sage_prepass_kv_varlen(const void *k, const void *v, void *k_mean, int8_t *k_int8, float *k_scale, void *v_image,
                                    float *ws, uint32_t *sync, const int32_t *cu_seqlens_k, const int32_t *cu_k_scale,
                                    const int32_t *slab_first, const int32_t *slab_seq, const int32_t *hdr,
                                    int nseq, int total_tokens, int max_seqlen_k, int nslab_bound, int H, int D,
                                    int64_t k_sl, int64_t k_sh, int64_t v_sl, int64_t v_sh, int64_t ko_sl, int64_t ko_sh,
                                    int dtype, uint32_t *host_flag, void *stream)
{
    SAGE_REQUIRE(k && k_int8 && k_scale, "the varlen pre-pass needs k, k_int8 and k_scale");
    SAGE_REQUIRE(!v || v_image, "V part needs v_image");
    SAGE_REQUIRE(ws && sync, "the fused pre-pass needs its workspace and its sync buffer");
    SAGE_REQUIRE(cu_seqlens_k && cu_k_scale && slab_first && slab_seq && hdr, "the varlen pre-pass needs the index arrays of sage_varlen_plan");
    SAGE_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128 (got %d)", D);
    SAGE_REQUIRE(nseq > 0 && H > 0 && total_tokens > 0 && max_seqlen_k > 0 && nslab_bound > 0, "empty batch");
    SAGE_REQUIRE(H <= 65535, "head count too large for one launch (%d)", H);
    SAGE_REQUIRE(nslab_bound >= (total_tokens + sage::kStatsSlab - 1) / sage::kStatsSlab, "nslab_bound (%d) is below ceil(total_tokens / %d)", nslab_bound, sage::kStatsSlab);
    SAGE_REQUIRE(dtype == SAGE_DTYPE_F16 || dtype == SAGE_DTYPE_BF16, "bad dtype %d", dtype);
    {   // every slab of a head (all sequences) waits for the others when the K mean is asked for: they must be co-resident
        int dev = 0;
        SAGE_REQUIRE(hipGetDevice(&dev) == hipSuccess, "no current device");
        const int cus = stream_cu_count(static_cast<hipStream_t>(stream), dev);
        SAGE_REQUIRE(k_mean == nullptr || (nslab_bound <= sage::kPrepassMaxSlabs && nslab_bound <= cus),
                     "up to %d slabs per head cannot wait for each other inside one launch (limit %d, %d compute units on this stream): use the "
                     "sage_channel_mean / sage_quant_qk_int8_varlen / sage_prep_v_f16_varlen sequence", nslab_bound, sage::kPrepassMaxSlabs, cus);
    }
    const int64_t lpad = ((int64_t)max_seqlen_k + sage::kStatsSlab - 1) / sage::kStatsSlab * sage::kStatsSlab;
    SAGE_REQUIRE(aligned16(k) && aligned16(k_int8) && (!v || (aligned16(v) && aligned16(v_image))), "k / k_int8 / v / v_image must be 16-byte aligned");
    SAGE_REQUIRE(k_sl % 8 == 0 && k_sh % 8 == 0 && (!v || (v_sl % 8 == 0 && v_sh % 8 == 0)), "input strides must be multiples of 8 elements");
    SAGE_REQUIRE(ko_sl % 16 == 0 && ko_sh % 16 == 0, "int8 output strides must be multiples of 16");
    SAGE_REQUIRE(((lpad - 1) * k_sl + D) * 2 < (int64_t)1 << 32 && (lpad - 1) * ko_sl + D < (int64_t)1 << 32 &&
                 (!v || ((lpad - 1) * v_sl + D) * 2 < (int64_t)1 << 32),
                 "one sequence of one head (rounded up to whole 512-row slabs) spans 4 GiB or more: the kernel addresses it with 32-bit buffer offsets");
    sage::PrepassParams p{};
    p.parts = 1 | (v ? 2 : 0);
    p.k = k; p.v = v; p.k_mean = k_mean; p.k_out = k_int8; p.k_scale = k_scale;
    p.v_image = v_image; p.v_scale = nullptr; p.v_mean = nullptr; p.ws = ws; p.sync = sync;
    p.B = 1; p.H = H; p.L = total_tokens; p.D = D; p.nslab = nslab_bound;
    p.k_sb = 0; p.k_sh = k_sh; p.k_sl = k_sl; p.v_sb = 0; p.v_sh = v_sh; p.v_sl = v_sl;
    p.ko_sb = 0; p.ko_sh = ko_sh; p.ko_sl = ko_sl;
    p.k_blk = 64; p.k_warp = 64; p.k_gran = sage::GR_BLOCK; p.k_style = sage::QS_TRITON;      // quant_per_block_varlen.py:21-58
    p.dtype = dtype; p.scale_max = 448.0f; p.v_fp16 = 1;
    p.debug_fail = g_prepass_debug_fail;
    p.host_flag = host_flag;
    p.cu = cu_seqlens_k; p.cu_tiles = cu_k_scale; p.slab_seq = slab_seq; p.slab_first = slab_first; p.hdr = hdr; p.nseq = nseq;
    return check_launch(sage::launch_prepass_kv(p, static_cast<hipStream_t>(stream)), "sage_prepass_kv_varlen launch");
}

// test hook: `nwg` workgroups of 1024 threads that spin for `ms` milliseconds (two of them fill a compute unit's wave slots)
SAGE_API int sage_debug_spin(int ms, int nwg, void *stream)
{
    SAGE_REQUIRE(ms > 0 && ms <= 10000 && nwg > 0 && nwg <= 4096, "sage_debug_spin: bad argument");
    return check_launch(sage::launch_debug_spin(ms, nwg, static_cast<hipStream_t>(stream)), "sage_debug_spin launch");
}

SAGE_API int sage_prep_v_f16(const void *v, void *v_image, const float *v_mean, int B, int H, int L, int D,
                    int64_t v_sb, int64_t v_sh, int64_t v_sl, int dtype, void *stream)
{
    return prep_v_common(v, v_image, nullptr, nullptr, v_mean, nullptr, nullptr, nullptr, B, H, L, D, v_sb, v_sh, v_sl,
                         0.0f, dtype, 0, stream);
}

SAGE_API int sage_prep_v_f16_varlen(const void *v, void *v_image, const int32_t *cu_seqlens, const int32_t *cu_tiles,
                           int nseq, int max_seqlen, int H, int D, int64_t v_sl, int64_t v_sh, int dtype, void *stream)
{
    SAGE_REQUIRE(cu_seqlens && cu_tiles, "varlen needs cu_seqlens and cu_tiles");
    return prep_v_common(v, v_image, nullptr, nullptr, nullptr, nullptr, cu_seqlens, cu_tiles, nseq, H, max_seqlen, D, 0, v_sh, v_sl,
                         0.0f, dtype, 0, stream);
}

SAGE_API int64_t sage_attn_launch_ws_bytes(void) { return sage::kAttnSchedBytes; }
SAGE_API int sage_attn_qk_int8_pv_f8(const int8_t *q, const int8_t *k, const void *v_image, void *o, float *lse,
                            const float *q_scale, const float *k_scale, const float *v_scale, const float *v_mean,
                            int B, int Hq, int Hkv, int Lq, int Lk, int D,
                            int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                            int64_t o_sb, int64_t o_sh, int64_t o_sl,
                            int is_causal, int qk_quant_gran, int q_warp,
                            float sm_scale_log2, int pv_accum, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    return attn_common(true, false, q, k, v_image, o, lse, q_scale, k_scale, v_scale, v_mean, nullptr, nullptr, nullptr, nullptr,
                       B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl,
                       is_causal, qk_quant_gran, q_warp, sm_scale_log2, pv_accum, out_dtype, stream, attr);
}

SAGE_API int sage_attn_qk_int8_pv_f16(const int8_t *q, const int8_t *k, const void *v_image, void *o, float *lse,
                             const float *q_scale, const float *k_scale, const float *v_mean,
                             int B, int Hq, int Hkv, int Lq, int Lk, int D,
                             int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                             int64_t o_sb, int64_t o_sh, int64_t o_sl,
                             int is_causal, int qk_quant_gran, int q_warp,
                             float sm_scale_log2, int pv_accum, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    return attn_common(false, false, q, k, v_image, o, lse, q_scale, k_scale, nullptr, v_mean, nullptr, nullptr, nullptr, nullptr,
                       B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl,
                       is_causal, qk_quant_gran, q_warp, sm_scale_log2, pv_accum, out_dtype, stream, attr);
}

SAGE_API int sage_attn_qk_int8_pv_f16_vrows(const int8_t *q, const int8_t *k, const void *v, void *o, float *lse,
                                   const float *q_scale, const float *k_scale, const float *v_mean,
                                   int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                   int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                   int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                   int is_causal, int qk_quant_gran, int q_warp,
                                   float sm_scale_log2, int pv_accum, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    const int64_t vs[3] = {v_sb, v_sh, v_sl};
    return attn_common(false, false, q, k, v, o, lse, q_scale, k_scale, nullptr, v_mean, nullptr, nullptr, nullptr, nullptr,
                       B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl,
                       is_causal, qk_quant_gran, q_warp, sm_scale_log2, pv_accum, out_dtype, stream, attr, nullptr, nullptr, nullptr, nullptr, 0, vs);
}

SAGE_API int sage_attn_qk_int8_pv_f16_masked(const int8_t *q, const int8_t *k, const void *v_image, void *o, float *lse,
                                    const float *q_scale, const float *k_scale, const void *mask, int mask_kind,
                                    int64_t m_sb, int64_t m_sh, int64_t m_sq, int64_t m_sk,
                                    int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                    int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                    int64_t o_sb, int64_t o_sh, int64_t o_sl, float sm_scale_log2, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    const MaskArg m{mask, mask_kind, m_sb, m_sh, m_sq, m_sk};
    return attn_common(false, false, q, k, v_image, o, lse, q_scale, k_scale, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                       B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl,
                       0, SAGE_GRAN_PER_BLOCK, 128, sm_scale_log2, SAGE_PV_ACCUM_TRITON, out_dtype, stream, attr, &m);
}

SAGE_API int sage_attn_qk_int8_pv_f16_varlen(const int8_t *q, const int8_t *k, const void *v_image, void *o,
                                    const float *q_scale, const float *k_scale,
                                    const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k,
                                    const int32_t *cu_q_scale, const int32_t *cu_k_scale, const int32_t *seq_order,
                                    const int32_t *work_items, const int32_t *work_hdr, int items_bound,
                                    int nseq, int max_seqlen_q, int Hq, int Hkv, int D,
                                    int64_t q_sl, int64_t q_sh, int64_t k_sl, int64_t k_sh, int64_t o_sl, int64_t o_sh,
                                    int is_causal, float sm_scale_log2, int pv_accum, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    return attn_common(false, true, q, k, v_image, o, nullptr, q_scale, k_scale, nullptr, nullptr,
                       cu_seqlens_q, cu_seqlens_k, cu_q_scale, cu_k_scale,
                       nseq, Hq, Hkv, max_seqlen_q, 0, D, 0, q_sh, q_sl, 0, k_sh, k_sl, 0, o_sh, o_sl,
                       is_causal, SAGE_GRAN_PER_BLOCK, 128, sm_scale_log2, pv_accum, out_dtype, stream, attr, nullptr, seq_order,
                       work_items, work_hdr, items_bound);
}

static int fused_q_common(const void *q, const int8_t *k, const void *v_image, void *o, float *lse,
                          const float *k_scale, const float *v_scale, const float *v_mean,
                          int B, int Hq, int Hkv, int Lq, int Lk, int D,
                          int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                          int64_t o_sb, int64_t o_sh, int64_t o_sl,
                          int is_causal, float sm_scale_log2, int q_dtype, int out_dtype, int kv_split, void *stream, const SageLaunchAttr *attr,
                          bool pv_fp8 = true, const int64_t *v_strides = nullptr)
{
    LaunchAttr la;
    if (const int rc = read_attr(attr, stream, kv_split <= 1, la)) return rc;       // (split launches take no launch workspace)
    if (v_strides != nullptr) {        // V rows in place: fp16 q / k / v tensors of one call
        SAGE_REQUIRE(!pv_fp8 && kv_split <= 1 && q_dtype == SAGE_DTYPE_F16, "V rows in place: FP16 PV on fp16 inputs, no split");
        SAGE_REQUIRE(v_strides[0] % 8 == 0 && v_strides[1] % 8 == 0 && v_strides[2] % 8 == 0 && v_strides[2] >= D, "v strides must be multiples of 8 elements (16-byte rows)");
        SAGE_REQUIRE(((int64_t)(Lk - 1) * v_strides[2] + D) * 2 < (int64_t)1 << 31, "one head of v must span less than 2 GiB");
    }
    SAGE_REQUIRE(q && k && v_image && o && k_scale && (v_scale || !pv_fp8), "null tensor pointer");
    SAGE_REQUIRE(kv_split >= 0 && (kv_split <= 1 || Hkv % kv_split == 0), "kv_split (%d) must divide the folded kv-head count (%d)", kv_split, Hkv);
    SAGE_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128 (got %d); pad on the host as core.py:260-271 does", D);
    SAGE_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Lq > 0 && Lk > 0, "empty problem (B=%d Hq=%d Hkv=%d Lq=%d Lk=%d)", B, Hq, Hkv, Lq, Lk);
    SAGE_REQUIRE(Hq % Hkv == 0, "num_qo_heads (%d) must be divisible by num_kv_heads (%d)", Hq, Hkv);
    SAGE_REQUIRE(q_dtype == SAGE_DTYPE_F16 || q_dtype == SAGE_DTYPE_BF16, "bad q_dtype %d", q_dtype);
    SAGE_REQUIRE(out_dtype == SAGE_DTYPE_F16 || out_dtype == SAGE_DTYPE_BF16, "bad out_dtype %d", out_dtype);
    SAGE_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v_image) && aligned16(o), "q/k/v/o must be 16-byte aligned");
    SAGE_REQUIRE(q_sl % 8 == 0 && q_sh % 8 == 0 && q_sb % 8 == 0, "q strides must be multiples of 8 elements");
    SAGE_REQUIRE(k_sl % 16 == 0 && k_sh % 16 == 0 && k_sb % 16 == 0, "int8 k strides must be multiples of 16");
    SAGE_REQUIRE(o_sl % 8 == 0 && o_sh % 8 == 0 && o_sb % 8 == 0, "output strides must be multiples of 8 elements");
    sage::AttnParams p{};
    p.sched = la.ws; p.trace = la.trace; p.trace_wgs = la.trace_wgs;
    p.q = q; p.k = k; p.v = v_image; p.o = o; p.lse = lse;
    p.k_scale = k_scale; p.v_scale = v_scale; p.v_mean = v_mean;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.group = Hq / Hkv;
    p.Lq = Lq; p.Lk = Lk;
    p.nqblk = (Lq + sage::BLKQ - 1) / sage::BLKQ;
    p.q_sb = q_sb; p.q_sh = q_sh; p.q_sl = q_sl;
    p.k_sb = k_sb; p.k_sh = k_sh; p.k_sl = k_sl;
    p.o_sb = o_sb; p.o_sh = o_sh; p.o_sl = o_sl;
    p.q_gran = sage::QG_PER_THREAD; p.qs_per_blk = 32;
    p.nqs = p.nqblk * p.qs_per_blk;
    p.nks = ((Lk + sage::BLKK - 1) / sage::BLKK) * 4;
    p.out_dtype = out_dtype;
    p.sm_scale_log2 = sm_scale_log2;
    p.kv_split = kv_split;
    if (v_strides != nullptr) { p.v_rows = 1; p.v_sb = v_strides[0]; p.v_sh = v_strides[1]; p.v_sl = v_strides[2]; }
    return check_launch(sage::launch_attn_fused_q(p, D, is_causal != 0, q_dtype, pv_fp8, la.opts), "sage_attn_fused_q launch");
}

SAGE_API int sage_attn_fused_q_pv_f8(const void *q, const int8_t *k, const void *v_image, void *o, float *lse,
                                     const float *k_scale, const float *v_scale, const float *v_mean,
                                     int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                     int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                     int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                     int is_causal, float sm_scale_log2, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    return fused_q_common(q, k, v_image, o, lse, k_scale, v_scale, v_mean, B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl,
                          o_sb, o_sh, o_sl, is_causal, sm_scale_log2, q_dtype, out_dtype, 0, stream, attr);
}

SAGE_API int sage_attn_fused_q_pv_f16(const void *q, const int8_t *k, const void *v_image, void *o, float *lse,
                                      const float *k_scale, const float *v_mean,
                                      int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                      int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                      int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                      int is_causal, float sm_scale_log2, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    return fused_q_common(q, k, v_image, o, lse, k_scale, nullptr, v_mean, B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl,
                          o_sb, o_sh, o_sl, is_causal, sm_scale_log2, q_dtype, out_dtype, 0, stream, attr, false);
}

SAGE_API int sage_attn_fused_q_pv_f16_vrows(const void *q, const int8_t *k, const void *v, void *o, float *lse, const float *k_scale,
                                            int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                            int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                            int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                            int is_causal, float sm_scale_log2, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    const int64_t vs[3] = {v_sb, v_sh, v_sl};
    return fused_q_common(q, k, v, o, lse, k_scale, nullptr, nullptr, B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl,
                          o_sb, o_sh, o_sl, is_causal, sm_scale_log2, SAGE_DTYPE_F16, out_dtype, 0, stream, attr, false, vs);
}

// q in fp16 / bf16, quantised per 128-row block in the kernel prologue (dense: cu_q == nullptr; varlen: packed tensors, B = nseq)
static int fused_qblock_common(const void *q, const int8_t *k, const void *v_image, void *o, float *lse, const float *k_scale,
                               const int32_t *cu_q, const int32_t *cu_k, const int32_t *cu_ks, const int32_t *seq_order,
                               const int32_t *work_items, const int32_t *work_hdr, int items_bound,
                               int B, int Hq, int Hkv, int Lq, int Lk, int D,
                               int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                               int64_t o_sb, int64_t o_sh, int64_t o_sl,
                               int is_causal, float q_premul, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr,
                               const int64_t *v_strides = nullptr)
{
    LaunchAttr la;
    if (const int rc = read_attr(attr, stream, true, la)) return rc;
    const bool varlen = cu_q != nullptr;
    if (v_strides != nullptr) {
        SAGE_REQUIRE(!varlen && q_dtype == SAGE_DTYPE_F16, "V rows in place: dense calls on fp16 inputs");
        SAGE_REQUIRE(v_strides[0] % 8 == 0 && v_strides[1] % 8 == 0 && v_strides[2] % 8 == 0 && v_strides[2] >= D, "v strides must be multiples of 8 elements (16-byte rows)");
        SAGE_REQUIRE(((int64_t)(Lk - 1) * v_strides[2] + D) * 2 < (int64_t)1 << 31, "one head of v must span less than 2 GiB");
    }
    SAGE_REQUIRE(q && k && v_image && o && k_scale, "null tensor pointer");
    SAGE_REQUIRE(D == 64 || D == 128, "head_dim must be 64 or 128 (got %d); pad on the host as core.py:260-271 does", D);
    SAGE_REQUIRE(B > 0 && Hq > 0 && Hkv > 0 && Lq > 0 && (varlen || Lk > 0), "empty problem (B=%d Hq=%d Hkv=%d Lq=%d Lk=%d)", B, Hq, Hkv, Lq, Lk);
    SAGE_REQUIRE(Hq % Hkv == 0, "num_qo_heads (%d) must be divisible by num_kv_heads (%d)", Hq, Hkv);
    SAGE_REQUIRE(q_dtype == SAGE_DTYPE_F16 || q_dtype == SAGE_DTYPE_BF16, "bad q_dtype %d", q_dtype);
    SAGE_REQUIRE(out_dtype == SAGE_DTYPE_F16 || out_dtype == SAGE_DTYPE_BF16, "bad out_dtype %d", out_dtype);
    SAGE_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v_image) && aligned16(o), "q/k/v/o must be 16-byte aligned");
    SAGE_REQUIRE(q_sl % 8 == 0 && q_sh % 8 == 0 && q_sb % 8 == 0, "q strides must be multiples of 8 elements");
    SAGE_REQUIRE(k_sl % 16 == 0 && k_sh % 16 == 0 && k_sb % 16 == 0, "int8 k strides must be multiples of 16");
    SAGE_REQUIRE(o_sl % 8 == 0 && o_sh % 8 == 0 && o_sb % 8 == 0, "output strides must be multiples of 8 elements");
    SAGE_REQUIRE(!varlen || (cu_k && cu_ks), "varlen needs cu_seqlens_k and the k scale prefix array");
    SAGE_REQUIRE(!varlen || lse == nullptr, "varlen returns no lse");
    sage::AttnParams p{};
    p.sched = la.ws; p.trace = la.trace; p.trace_wgs = la.trace_wgs;
    p.q = q; p.k = k; p.v = v_image; p.o = o; p.lse = lse;
    p.k_scale = k_scale;
    p.cu_q = cu_q; p.cu_k = cu_k; p.cu_qs = nullptr; p.cu_ks = cu_ks; p.seq_order = varlen ? seq_order : nullptr;
    SAGE_REQUIRE((work_items == nullptr) == (work_hdr == nullptr) && (work_items == nullptr || (varlen && items_bound > 0)),
                 "the work list comes as (work_items, work_hdr, items_bound > 0), varlen only");
    p.work_items = work_items; p.work_hdr = work_hdr; p.items_bound = items_bound;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.group = Hq / Hkv;
    p.Lq = Lq; p.Lk = Lk;
    p.nqblk = (Lq + sage::BLKQ - 1) / sage::BLKQ;
    p.q_sb = q_sb; p.q_sh = q_sh; p.q_sl = q_sl;
    p.k_sb = k_sb; p.k_sh = k_sh; p.k_sl = k_sl;
    p.o_sb = o_sb; p.o_sh = o_sh; p.o_sl = o_sl;
    p.q_gran = sage::QG_PER_BLOCK; p.qs_per_blk = 1;
    p.nqs = p.nqblk;
    p.nks = (Lk + sage::BLKK - 1) / sage::BLKK;
    p.out_dtype = out_dtype;
    p.sm_scale_log2 = 1.0f;                 // sm_scale * log2(e) is folded into the quantised q (q_premul), as the reference's quantiser does
    p.q_premul = q_premul;
    if (v_strides != nullptr) { p.v_rows = 1; p.v_sb = v_strides[0]; p.v_sh = v_strides[1]; p.v_sl = v_strides[2]; }
    return check_launch(sage::launch_attn_fused_qblock(p, D, is_causal != 0, q_dtype, la.opts), "sage_attn_fused_qblock launch");
}

SAGE_API int sage_attn_fused_qblock_pv_f16(const void *q, const int8_t *k, const void *v_image, void *o, float *lse, const float *k_scale,
                                           int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                           int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                           int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                           int is_causal, float q_premul, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    return fused_qblock_common(q, k, v_image, o, lse, k_scale, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, B, Hq, Hkv, Lq, Lk, D,
                               q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl, is_causal, q_premul, q_dtype, out_dtype, stream, attr);
}

SAGE_API int sage_attn_fused_qblock_pv_f16_vrows(const void *q, const int8_t *k, const void *v, void *o, float *lse, const float *k_scale,
                                                 int B, int Hq, int Hkv, int Lq, int Lk, int D,
                                                 int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                                 int64_t v_sb, int64_t v_sh, int64_t v_sl, int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                                 int is_causal, float q_premul, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    const int64_t vs[3] = {v_sb, v_sh, v_sl};
    return fused_qblock_common(q, k, v, o, lse, k_scale, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, B, Hq, Hkv, Lq, Lk, D,
                               q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl, is_causal, q_premul, SAGE_DTYPE_F16, out_dtype, stream, attr, vs);
}

SAGE_API int sage_attn_fused_qblock_pv_f16_varlen(const void *q, const int8_t *k, const void *v_image, void *o, const float *k_scale,
                                                  const int32_t *cu_seqlens_q, const int32_t *cu_seqlens_k, const int32_t *cu_k_scale,
                                                  const int32_t *seq_order, const int32_t *work_items, const int32_t *work_hdr, int items_bound,
                                                  int nseq, int max_seqlen_q, int Hq, int Hkv, int D,
                                                  int64_t q_sl, int64_t q_sh, int64_t k_sl, int64_t k_sh, int64_t o_sl, int64_t o_sh,
                                                  int is_causal, float q_premul, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    SAGE_REQUIRE(cu_seqlens_q != nullptr, "varlen needs cu_seqlens_q");
    return fused_qblock_common(q, k, v_image, o, nullptr, k_scale, cu_seqlens_q, cu_seqlens_k, cu_k_scale, seq_order,
                               work_items, work_hdr, items_bound, nseq, Hq, Hkv, max_seqlen_q, 0, D, 0, q_sh, q_sl, 0, k_sh, k_sl, 0, o_sh, o_sl,
                               is_causal, q_premul, q_dtype, out_dtype, stream, attr);
}

SAGE_API int sage_attn_fused_q_pv_f8_split(const void *q, const int8_t *k, const void *v_image, void *o_part, float *lse_part,
                                           const float *k_scale, const float *v_scale, const float *v_mean,
                                           int B, int Hq, int Hkv, int kv_split, int Lq, int Lk_chunk, int D,
                                           int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                           int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                           int is_causal, float sm_scale_log2, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    SAGE_REQUIRE(kv_split >= 2, "kv_split must be at least 2 (got %d)", kv_split);
    SAGE_REQUIRE(o_part && lse_part, "split-KV needs the partial output and log-sum-exp buffers");
    SAGE_REQUIRE(Lk_chunk % 64 == 0, "split-KV chunks are whole numbers of 64-key tiles (got %d keys)", Lk_chunk);
    return fused_q_common(q, k, v_image, o_part, lse_part, k_scale, v_scale, v_mean, B, Hq * kv_split, Hkv * kv_split, Lq, Lk_chunk, D,
                          q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl, is_causal, sm_scale_log2, q_dtype, out_dtype, kv_split, stream, attr);
}

SAGE_API int sage_attn_fused_q_pv_f16_split(const void *q, const int8_t *k, const void *v_image, void *o_part, float *lse_part,
                                            const float *k_scale, const float *v_mean,
                                            int B, int Hq, int Hkv, int kv_split, int Lq, int Lk_chunk, int D,
                                            int64_t q_sb, int64_t q_sh, int64_t q_sl, int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                            int64_t o_sb, int64_t o_sh, int64_t o_sl,
                                            int is_causal, float sm_scale_log2, int q_dtype, int out_dtype, void *stream, const SageLaunchAttr *attr)
{
    SAGE_REQUIRE(kv_split >= 2, "kv_split must be at least 2 (got %d)", kv_split);
    SAGE_REQUIRE(o_part && lse_part, "split-KV needs the partial output and log-sum-exp buffers");
    SAGE_REQUIRE(Lk_chunk % 64 == 0, "split-KV chunks are whole numbers of 64-key tiles (got %d keys)", Lk_chunk);
    return fused_q_common(q, k, v_image, o_part, lse_part, k_scale, nullptr, v_mean, B, Hq * kv_split, Hkv * kv_split, Lq, Lk_chunk, D,
                          q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl, is_causal, sm_scale_log2, q_dtype, out_dtype, kv_split, stream,
                          attr, false);
}

SAGE_API int sage_merge_states(float *o_acc, float *lse_acc, const void *o_new, const float *lse_new, void *o_out,
                               int B, int H, int L, int D, int64_t n_sb, int64_t n_sh, int64_t n_sl,
                               int64_t o_sb, int64_t o_sh, int64_t o_sl, int dtype, int first, void *stream)
{
    SAGE_REQUIRE(o_acc && lse_acc && o_new && lse_new, "null tensor pointer");
    SAGE_REQUIRE(B > 0 && H > 0 && L > 0, "empty problem (B=%d H=%d L=%d)", B, H, L);
    SAGE_REQUIRE(D > 0 && D % 8 == 0 && D <= 512, "head_dim must be a positive multiple of 8, at most 512 (got %d)", D);
    SAGE_REQUIRE(dtype == SAGE_DTYPE_F16 || dtype == SAGE_DTYPE_BF16, "bad dtype %d", dtype);
    SAGE_REQUIRE(aligned16(o_acc) && aligned16(o_new) && (o_out == nullptr || aligned16(o_out)), "o tensors must be 16-byte aligned");
    SAGE_REQUIRE(n_sb % 8 == 0 && n_sh % 8 == 0 && n_sl % 8 == 0, "o_new strides must be multiples of 8 elements");
    SAGE_REQUIRE(o_out == nullptr || (o_sb % 8 == 0 && o_sh % 8 == 0 && o_sl % 8 == 0), "o_out strides must be multiples of 8 elements");
    sage::MergeParams p{};
    p.o_acc = o_acc; p.lse_acc = lse_acc; p.o_new = o_new; p.lse_new = lse_new; p.o_out = o_out;
    p.B = B; p.H = H; p.L = L; p.D = D;
    p.n_sb = n_sb; p.n_sh = n_sh; p.n_sl = n_sl; p.o_sb = o_sb; p.o_sh = o_sh; p.o_sl = o_sl;
    p.dtype = dtype; p.first = first != 0;
    const hipError_t e = sage::launch_merge_states(p, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(SAGE_ELAUNCH, "sage_merge_states launch: %s", hipGetErrorString(e));
    return SAGE_OK;
}

SAGE_API int sage_merge_split(const void *o_part, const float *lse_part, const void *o_tail, const float *lse_tail,
                              void *o_out, float *lse_out, int B, int S, int H, int group, int L, int D,
                              int64_t o_sb, int64_t o_sh, int64_t o_sl, int out_dtype, void *stream)
{
    SAGE_REQUIRE(group > 0 && H % group == 0, "num heads (%d) must be divisible by the GQA group size (%d)", H, group);
    SAGE_REQUIRE(o_part && lse_part && o_out, "null tensor pointer");
    SAGE_REQUIRE((o_tail == nullptr) == (lse_tail == nullptr), "o_tail and lse_tail come together");
    SAGE_REQUIRE(B > 0 && S > 0 && H > 0 && L > 0, "empty problem (B=%d S=%d H=%d L=%d)", B, S, H, L);
    SAGE_REQUIRE(D > 0 && D % 8 == 0 && D <= 512, "head_dim must be a positive multiple of 8, at most 512 (got %d)", D);
    SAGE_REQUIRE(out_dtype == SAGE_DTYPE_F16 || out_dtype == SAGE_DTYPE_BF16, "bad out_dtype %d", out_dtype);
    SAGE_REQUIRE(aligned16(o_part) && aligned16(o_out) && (o_tail == nullptr || aligned16(o_tail)), "o tensors must be 16-byte aligned");
    SAGE_REQUIRE(o_sb % 8 == 0 && o_sh % 8 == 0 && o_sl % 8 == 0, "o_out strides must be multiples of 8 elements");
    sage::SplitMergeParams p{};
    p.o_part = o_part; p.lse_part = lse_part; p.o_tail = o_tail; p.lse_tail = lse_tail; p.o_out = o_out; p.lse_out = lse_out;
    p.B = B; p.S = S; p.H = H; p.L = L; p.D = D; p.group = group; p.o_sb = o_sb; p.o_sh = o_sh; p.o_sl = o_sl; p.dtype = out_dtype;
    const hipError_t e = sage::launch_merge_split(p, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(SAGE_ELAUNCH, "sage_merge_split launch: %s", hipGetErrorString(e));
    return SAGE_OK;
}

}  // extern "C"
