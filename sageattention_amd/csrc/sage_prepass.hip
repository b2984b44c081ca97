// sage_prepass.hip -- the K / V pre-pass of one attention call as ONE launch that reads K and V once.
//
// What it replaces (same bits, see tests/test_gpu_prepass.py):
//   K: stats_partial + stats_final (k.mean(dim=seq), core.py:280) + quant_int8_kernel (fused.cu:64-198 /
//      quant_per_thread.py:58-98 with the `k - km` of core.py:281 fused)                       3 launches, 4 B/elt read
//   V: stats_partial + stats_final + prep_v_kernel (fused.cu:262-427, quant.py:224-293)         3 launches, 4 B/elt read
// A 512-thread workgroup owns one 512-token slab of one (batch, kv-head) of K or of V and keeps it in REGISTERS (32 rows x
// 8 B per thread at D = 128) across the three steps
//   1. per-channel statistics of the slab -> workspace: sums for K, (max, min, sum) for V   (summation order of sage_stats.hip)
//   2. wait until the other slabs of the head have published theirs; reduce them in slab order
//   3. quantise the slab from the registers: INT8 rows + group scales (K), FP8 PV-operand tile image (V)
// so HBM sees 2 B/elt in and 1 B/elt out -- the algorithmic minimum -- instead of 4 + 1.
//
// Step 2 is a per-head barrier between workgroups of one launch.  It is safe because (a) gfx950 hands workgroups to
// its 8 XCDs round-robin and each XCD starts its share in index order, (b) the slabs of a head are consecutive in
// index, so the lowest unfinished head always has every slab resident or next in line, and (c) the C ABI refuses
// heads of more than kPrepassMaxSlabs slabs (16 per XCD) or than the device has CUs, far below the resident-workgroup
// capacity (2 per CU).
// The same assumption carries rocPRIM's decoupled look-back scan.  Cross-XCD visibility of the partials follows the
// gfx942+ memory model for atomics: the arrival counter and the partials are agent-scope atomic accesses.
// The two counters of a head return to zero before the kernel ends -- the last slab to leave re-arms them, also in a launch in which a
// workgroup gave up -- so ONE buffer that its owner zeroed ONCE serves every call issued in stream order: there is no zeroing launch in
// front of the kernel any more (round 5: that launch was 4.8 us plus a kernel boundary of every sageattn() call).  The wait is bounded
// (30 ms of the 100 MHz wall clock) and the assumption is not load-bearing for correctness: a workgroup whose head-mates have not all arrived in time sets
// word 2 of the head's sync line (and the caller's pinned host word), then computes the head's statistics ITSELF -- it streams the whole
// head through a rolled copy of the statistics pass, slab by slab in index order, which reproduces every slab's partial bit for bit --
// and carries on.  The launch is then slow (nslab x the read traffic for that workgroup), never wrong: rounds 2-3 poisoned the outputs
// with NaN instead.  sage_prepass_failed_heads() reads the flags back; the Python layer routes the device's later calls through the kernel
// sequence once its host word is set (a performance decision, quant._PrepassGuard).
#include "sage_common.h"
#include "sage_kernels.h"
#include "sage_quant_math.h"
#include <type_traits>

#ifndef SAGE_PP_RELEASE
#define SAGE_PP_RELEASE 0  // experiment: 1 = the arrival is an agent-scope RELEASE (adds one buffer_wbl2 per workgroup)
#endif
#ifndef SAGE_PP_NT
#define SAGE_PP_NT 1       // non-temporal stores of the INT8 / FP8 output (written once, read by a later kernel): 123 -> 99 us
                           // at C3 in back-to-back launches, neutral inside sageattn(); 0 = ordinary stores
#endif
#ifndef SAGE_PP_TRACE
#define SAGE_PP_TRACE 0    // experiment: thread 0 of every workgroup appends 100 MHz time stamps behind the used part of ws
#endif
#ifndef SAGE_PP_KWB
#define SAGE_PP_KWB 1      // K, Triton rounding styles: the abs-max pass writes the smoothed, rounded pair back over the raw word
#endif
#ifndef SAGE_PP_VAMAX
#define SAGE_PP_VAMAX 1    // V without smooth_v: statistics = abs-max on the 16-bit patterns
#endif
#ifndef SAGE_PP_ABL
#define SAGE_PP_ABL 0      // experiment bits (wrong results): 1 no wait, 2 no quantise step, 4 no slab statistics
#endif

namespace sage {

// `x - mean` rounded to the input dtype and widened again (torch's `k - km`); gfx950 rounds a pair of fp32 to bf16 in one
// instruction (v_cvt_pk_bf16_f32, round-to-nearest-even like f32_to_bf16_rne)
template <int DT> __device__ __forceinline__ void round_pair_to_dtype(float &a, float &b)
{
    if constexpr (DT == DT_BF16) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const b2 r = __builtin_convertvector(f2{a, b}, b2);
        unsigned u;
        __builtin_memcpy(&u, &r, 4);
        a = __uint_as_float(u << 16);
        b = __uint_as_float(u & 0xffff0000u);
    } else {
        a = f16_to_f32(f32_to_f16_rne(a));
        b = f16_to_f32(f32_to_f16_rne(b));
    }
}

constexpr int kPrepassThreads = 512;

// the same rounding, left packed (low half = a)
template <int DT> __device__ __forceinline__ unsigned pack_pair_to_dtype(float a, float b)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    unsigned u;
    if constexpr (DT == DT_BF16) {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const b2 r = __builtin_convertvector(f2{a, b}, b2);
        __builtin_memcpy(&u, &r, 4);
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 r = __builtin_convertvector(f2{a, b}, h2);
        __builtin_memcpy(&u, &r, 4);
    }
    return u;
}

template <int D>
struct PrepassLds {
    static constexpr int TPR = D / 4, RPI = kPrepassThreads / TPR, LDT = D + 8;
    float red[3][RPI][D];
    __attribute__((aligned(16))) uint16_t tile[2][BLKK * LDT];
    float ch_mean[D], ch_recp[D];
    uint16_t kmean[D];
    unsigned failed;          // the head's give-up flag as thread 0 found it after the wait
    unsigned gmax[8][8];
    float gsc[8][8], gy[8][8];
};

// The K half and the V half are two instantiations of this body under one workgroup-uniform branch of the kernel.  Written as
// one body with `is_v` selects they needed > 128 VGPRs and spilled (every reload is a memory round trip behind
// `s_waitcnt vmcnt(0)`: the passes ran 2.5x slower); as two disjoint regions they take 115 / 127 and nothing spills.
// VARLEN: k / v are packed [sum L, H, D] (sageattn_varlen, core.py:431-444): blockIdx.x is the slab's index among ALL slabs of the head
// (sage_varlen_plan: 512-token slabs per sequence, so the 64-key scale blocks and V tiles of a sequence never straddle a slab); the
// statistics, the barrier and the K mean span every sequence (`k.mean(dim=0)` over all packed tokens), the quantisation is per sequence.
template <int D, int DT, bool IS_V, bool VARLEN, bool V_AMAX = false>
__device__ __forceinline__ void prepass_body(const PrepassParams &p, PrepassLds<D> &lds, const int b)
{

    // 512 threads, 4 channels (8 B) per thread and row: a row is D / 4 threads wide, the workgroup passes over RPI rows at
    // a time and thread (r0, c4) owns rows r0 + RPI * i -- the row -> thread map of stats_partial_kernel (RPI = 16 at D = 128,
    // 32 at D = 64), so the per-channel sums associate the same way.  64 data VGPRs per thread (D = 128) leave room for
    // 4 waves / SIMD: two slabs per CU, sixteen waves to hide the latency of the compute steps.
    // Every loop over the rows is unrolled with static register indices; the code stays at ~5k instructions because a
    // thread has only 4 channels, the arithmetic is packed (v_pk_*_f32) and the rounding style is a template argument.
    // (History, C3 shape, six-launch sequence = 154 us: 8 channels x 256 threads fully unrolled with run-time style
    //  branches = 20k instructions, instruction-fetch bound, 437 us; rolled loops over s_set_gpr_idx-indexed register
    //  tuples = 177 us, ~100 cycles per element in the passes -- profiles/r2_run_r3e_trace_c3.txt.)
    constexpr int NT = kPrepassThreads;
    constexpr int TPR = D / 4;                  // threads per row
    constexpr int RPI = NT / TPR;               // rows per pass of the workgroup
    constexpr int NR = kStatsSlab / RPI;        // rows per thread: 32 (D = 128) / 16 (D = 64)
    constexpr int LDT = D + 8;
    auto &red = lds.red; auto &tile = lds.tile; auto &ch_mean = lds.ch_mean; auto &ch_recp = lds.ch_recp;
    auto &kmean = lds.kmean; auto &gmax = lds.gmax; auto &gsc = lds.gsc; auto &gy = lds.gy;
    bool failed = false;                        // this workgroup stopped waiting for its head-mates: it computes the head's statistics itself
    constexpr int is_v = IS_V ? 1 : 0;

    const int tid = threadIdx.x;
    const int gslab = blockIdx.x, h = blockIdx.y;          // gslab: index among the head's slabs (the statistics / barrier index)
    int slab = gslab, L = p.L, nslab = p.nslab;            // slab: index inside the sequence; L: rows of the sequence; nslab: slabs of the head
    long tok0 = 0, tile0 = 0;                              // VARLEN: first packed token / first 64-key block of the sequence
    bool gap = false;
    if constexpr (VARLEN) {
        typedef const __attribute__((address_space(4))) int *cint_p;          // wave-uniform: scalar loads
        nslab = ((cint_p)p.hdr)[4];
        if (gslab >= nslab) return;                        // the grid is sized by a host-known bound (workgroup-uniform exit)
        const int seg = ((cint_p)p.slab_seq)[gslab];
        slab = gslab - ((cint_p)p.slab_first)[seg];
        int t0, len;
        varlen_segment((cint_p)p.cu, p.nseq, p.L, seg, t0, len);
        tok0 = t0;
        L = len;
        gap = seg >= p.nseq;                                // rows outside every sequence: they count in the K mean and nowhere else
        if (gap && (IS_V || p.k_mean == nullptr)) return;
        tile0 = gap ? 0 : ((cint_p)p.cu_tiles)[seg];
    }
    const long bh = (long)b * p.H + h;
    const long x_sl = is_v ? p.v_sl : p.k_sl;
    const uint16_t *x = (is_v ? reinterpret_cast<const uint16_t *>(p.v) + (long)b * p.v_sb + (long)h * p.v_sh
                              : reinterpret_cast<const uint16_t *>(p.k) + (long)b * p.k_sb + (long)h * p.k_sh) + tok0 * x_sl;
    const int c4 = (tid % TPR) * 4, r0 = tid / TPR;
    const int row0 = slab * kStatsSlab;
    const int end = min(L, row0 + kStatsSlab);
    const int my_rows = (end - row0 - r0 + RPI - 1) / RPI;      // rows i < my_rows of this thread exist (may be <= 0)
#if SAGE_PP_TRACE
    unsigned long long *trace = reinterpret_cast<unsigned long long *>(p.ws + 2L * p.B * p.H * p.nslab * 3 * D) +
                                8L * (blockIdx.x + gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z));
    int tslot = 0;
#define SAGE_STAMP() do { if (tid == 0) trace[tslot] = wall_clock64(); tslot++; } while (0)
    SAGE_STAMP();
#else
#define SAGE_STAMP() do { } while (0)
#endif

    // ---- the slab, one read -------------------------------------------------------------------------------------
    unsigned rw[NR][2];
    {
        // buffer loads: rows past the end of the head read as zero (range check of the byte offset against num_records), so
        // the 32 loads need no branch -- a full / partial branch here costs a second copy of the slab in registers
        const unsigned head_bytes = (unsigned)(((long)(L - 1) * x_sl + D) * 2);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(x), 0, head_bytes, 0x00020000);
        const unsigned toff = (unsigned)((row0 + r0) * (int)x_sl + c4) * 2u;
        const unsigned step = (unsigned)(RPI * (int)x_sl) * 2u;
        unsigned voff = toff;                  // one running offset register (not 32 precomputed ones)
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const v2u t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0);
            rw[i][0] = t[0]; rw[i][1] = t[1];
            voff += step;
            asm volatile("" : "+v"(voff));
        }
    }
#if SAGE_PP_TRACE
    { unsigned acc = 0;
#pragma unroll
      for (int i = 0; i < NR; i++) acc += rw[i][0] ^ rw[i][1];
      if (acc == 0x12345u) p.ws[0] = 1.0f; }      // the loads have landed
    SAGE_STAMP();                                  // 1: slab loaded
#endif

    const bool need_stats = IS_V ? (p.v_fp16 == 0) : (p.k_mean != nullptr);      // the fp16 image needs no statistics, no barrier
    if (need_stats) {
        // ---- 1. slab statistics, rows in index order (bit-compatible with stats_partial_kernel) ----------------------
        float mx[4], mn[4], sm[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { mx[j] = -INFINITY; mn[j] = INFINITY; sm[j] = 0.0f; }
        // Rows past the end read as zeros: exact for the sums.  K only needs the sums (its mean).  For V a per-row penalty
        // (0 for a row that exists, -inf otherwise) is added for the max and subtracted for the min -- two full-rate adds
        // where a select per value (v_cndmask_b32: ~22 cycles per wave on gfx950, profiles/r2_run1_ubench2.txt) cost 4x more.
        // V without smooth_v (the default): the scale needs max |v| only -- abs-max on the 16-bit patterns (sign-magnitude formats order
        // like unsigned integers), one v_and + one v_pk_max_u16 per PAIR instead of six fp32 operations per value; rows past the end
        // read as zero and cannot win.  Published as (max, min) = (amax, 0): the reduction's max(|max - 0|, |min - 0|) is amax.
        // (a template argument, chosen by the kernel's workgroup-uniform branch: as a run-time branch inside one body the two forms of
        //  the statistics pass cost the bf16 D = 64 instantiation its third workgroup per CU -- 80 -> 93 VGPRs, 204 -> 235 us at C5)
        if constexpr (V_AMAX) {
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            us2 am01 = {0, 0}, am23 = {0, 0};
#pragma unroll
            for (int i = 0; i < ((SAGE_PP_ABL & 4) ? 0 : NR); i++) {
                const unsigned m0 = rw[i][0] & 0x7fff7fffu, m1 = rw[i][1] & 0x7fff7fffu;
                us2 a, b;
                __builtin_memcpy(&a, &m0, 4);
                __builtin_memcpy(&b, &m1, 4);
                am01 = __builtin_elementwise_max(am01, a);
                am23 = __builtin_elementwise_max(am23, b);
                asm volatile("" : "+v"(am01), "+v"(am23) :: "memory");      // keep the rows in order (tie the accumulators)
            }
            mx[0] = ld16<DT>(am01[0]); mx[1] = ld16<DT>(am01[1]); mx[2] = ld16<DT>(am23[0]); mx[3] = ld16<DT>(am23[1]);
#pragma unroll
            for (int j = 0; j < 4; j++) mn[j] = 0.0f;
        } else if (!(SAGE_PP_ABL & 4)) {
#pragma unroll
            for (int i = 0; i < NR; i++) {
                float pen = 0.0f;
                if constexpr (IS_V) pen = __uint_as_float((unsigned)((my_rows - 1 - i) >> 31) & 0xff800000u);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const unsigned w = rw[i][c];
                    const float lo = ld16<DT>((uint16_t)(w & 0xffffu)), hi = ld16<DT>((uint16_t)(w >> 16));
                    if constexpr (IS_V) {
                        mx[2 * c] = fmaxf(mx[2 * c], lo + pen);         mn[2 * c] = fminf(mn[2 * c], lo - pen);
                        mx[2 * c + 1] = fmaxf(mx[2 * c + 1], hi + pen); mn[2 * c + 1] = fminf(mn[2 * c + 1], hi - pen);
                    }
                    sm[2 * c] += lo;
                    sm[2 * c + 1] += hi;
                }
                // the rows stay in order (a "memory" clobber does not order register arithmetic; tying the accumulators does): the
                // compiler otherwise unpacks several rows ahead, and at D = 64 (80 VGPRs for three workgroups per CU) that is a spill
                asm volatile("" : "+v"(sm[0]), "+v"(sm[1]), "+v"(sm[2]), "+v"(sm[3]) :: "memory");
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { red[0][r0][c4 + j] = mx[j]; red[1][r0][c4 + j] = mn[j]; red[2][r0][c4 + j] = sm[j]; }
        __syncthreads();
        float *ws = p.ws + (((long)is_v * p.B * p.H + bh) * p.nslab) * 3 * D;
        if (tid < D) {
            float a = -INFINITY, c = INFINITY, s = 0.0f;
#pragma unroll
            for (int r = 0; r < RPI; r++) { a = fmaxf(a, red[0][r][tid]); c = fminf(c, red[1][r][tid]); s += red[2][r][tid]; }
            float *mine = ws + (long)gslab * 3 * D;
            // (what the head's reduction below reads: K the sums; V the maxima, and the minima and sums unless the abs-max suffices)
            if constexpr (IS_V) __hip_atomic_store(mine + tid, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (IS_V && !V_AMAX) __hip_atomic_store(mine + D + tid, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (!V_AMAX) __hip_atomic_store(mine + 2 * D + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SAGE_STAMP();                              // 2: slab statistics published
        // ---- 2. per-head barrier over the slabs, then the reduction in slab order (as stats_final_kernel) -------------
        unsigned *cnt = p.sync + ((long)is_v * p.B * p.H + bh) * kPrepassSyncStride;   // one 128-B line per head
        if (nslab > 1 && !(SAGE_PP_ABL & 1)) {
            // Everything that crosses workgroups here is an agent-scope atomic access (write-through / cache-bypassing,
            // coherent across the XCDs by itself), so the ordering needs no L2 write-back or invalidate -- a fence at
            // agent scope would put `buffer_wbl2` / `buffer_inv` into every spin iteration of every waiting workgroup
            // (measured: 2.3x slower).  Partials first, then the arrival: the stores are acknowledged (vmcnt) before
            // thread 0 counts the slab in; the readers issue their loads after the spin.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(cnt, 1u, SAGE_PP_RELEASE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // bounded (tens of milliseconds): if the forward-progress assumption above is violated -- compute units held by another
                // stream's kernels -- this workgroup stops waiting, records it (word 2 of the head's sync line, the caller's host word) and
                // computes the head's statistics ITSELF below; nothing it writes depends on another workgroup then
                const unsigned want = (unsigned)nslab + (p.debug_fail ? 1u : 0u);
                // a TIME bound (the 100 MHz wall clock: 30 ms; the test hook 0.2 ms) AND a minimum of polls.  Time, because how long a poll takes
                // depends on how contended the coherence point is and the bound is what separates "a slow head-mate" from "a head-mate that is
                // not running"; polls as well, because the wall clock keeps running while this wave is context-switched out (another process's
                // time slice on a shared device): a wave that comes back after 30 ms has not WAITED 30 ms, and its head-mates come back with it
                // (seen once: a give-up in the two-stream soak on a box that ran the suite 60 % slower than its neighbours)
                const long long bound = p.debug_fail ? 20000LL : 3000000LL;
                const unsigned min_polls = p.debug_fail ? (1u << 8) : (1u << 14);
                const long long t0 = wall_clock64();
                unsigned gave_up = 0, polls = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++polls > min_polls && wall_clock64() - t0 > bound) {
                        __hip_atomic_store(cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (p.host_flag != nullptr) __hip_atomic_store(p.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        gave_up = 1;
                        break;
                    }
                }
                lds.failed = gave_up;      // (a workgroup that arrives after another one gave up finds the count complete: the partials are all there)
            }
            __syncthreads();
            failed = lds.failed != 0;
        }
        SAGE_STAMP();                              // 3: every slab of the head has arrived
        float a = -INFINITY, c = INFINITY, s = 0.0f;
        if (failed) {
            // The head's other slabs did not all arrive in time.  Stream the whole head (every sequence's slabs of this kv head when packed)
            // through a rolled, register-light copy of the statistics pass, slab by slab in index order: each slab's partial is bit for bit
            // the one its own workgroup publishes (a thread's rows in index order, the threads of a channel in row order, the slabs in
            // index order), so the result does not depend on who computed it.  nslab x the read traffic of a workgroup, on a path that
            // would otherwise have produced garbage; the slab in rw is untouched.
            for (int i = 0; i < nslab; i++) {
                const uint16_t *xi = x;
                int Li = L, si = i;
                if constexpr (VARLEN) {
                    typedef const __attribute__((address_space(4))) int *cint_p;
                    const int seg = ((cint_p)p.slab_seq)[i];
                    si = i - ((cint_p)p.slab_first)[seg];
                    int t0;
                    varlen_segment((cint_p)p.cu, p.nseq, p.L, seg, t0, Li);
                    xi = x + ((long)t0 - tok0) * x_sl;
                }
                const int rbeg = si * kStatsSlab + r0, rend = min(Li, si * kStatsSlab + kStatsSlab);
                float fx[4], fn[4], fs[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { fx[j] = V_AMAX ? 0.0f : -INFINITY; fn[j] = V_AMAX ? 0.0f : INFINITY; fs[j] = 0.0f; }
                const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(xi), 0,
                                                                                      (unsigned)(((long)(Li - 1) * x_sl + D) * 2), 0x00020000);
                unsigned vo = (unsigned)(rbeg * (int)x_sl + c4) * 2u;
#pragma nounroll
                for (int r = rbeg; r < rend; r += RPI) {
                    const v2u t = __builtin_amdgcn_raw_buffer_load_b64(rs_i, vo, 0, 0);
                    vo += (unsigned)(RPI * (int)x_sl) * 2u;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const unsigned w = t[j >> 1];
                        const float f = ld16<DT>((uint16_t)((j & 1) ? (w >> 16) : (w & 0xffffu)));
                        if constexpr (V_AMAX) fx[j] = fmaxf(fx[j], fabsf(f));
                        else if constexpr (IS_V) { fx[j] = fmaxf(fx[j], f); fn[j] = fminf(fn[j], f); fs[j] += f; }
                        else fs[j] += f;                     // (K: the mean needs the sums only)
                    }
                }
                __syncthreads();                     // (the LDS scratch of the previous slab's reduction is free again)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if constexpr (IS_V) red[0][r0][c4 + j] = fx[j];
                    if constexpr (IS_V && !V_AMAX) red[1][r0][c4 + j] = fn[j];
                    if constexpr (!V_AMAX) red[2][r0][c4 + j] = fs[j];
                }
                __syncthreads();
                if (tid < D) {
                    float ai = -INFINITY, ci = INFINITY, s_i = 0.0f;
#pragma unroll
                    for (int r = 0; r < RPI; r++) {
                        if constexpr (IS_V) ai = fmaxf(ai, red[0][r][tid]);
                        if constexpr (IS_V && !V_AMAX) ci = fminf(ci, red[1][r][tid]);
                        if constexpr (!V_AMAX) s_i += red[2][r][tid];
                    }
                    if constexpr (IS_V) a = fmaxf(a, ai);
                    if constexpr (IS_V && !V_AMAX) c = fminf(c, ci);
                    if constexpr (!V_AMAX) s += s_i;
                }
            }
        } else if (tid < D) {
            // slabs in index order; sixteen slabs' loads are in flight together (one round trip to the coherence point per batch,
            // not one per slab: with the loads issued one by one this loop alone cost ~2 us x nslab per workgroup)
            constexpr int NB = (D == 128) ? 16 : 8;       // slabs per batch of loads (D = 64: 8 keeps the kernel at 3 workgroups per CU)
            for (int i0 = 0; i0 < nslab; i0 += NB) {
                float va[NB], vc[NB], vs[NB];
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const float *wi = ws + (long)min(i0 + u, nslab - 1) * 3 * D;
                    // (V with the abs-max only: one load per slab, not three -- the three cost a V workgroup 5.2 us of its 19.4 at C3 and
                    //  17 us of 37 on heads of 64 slabs, against 0.85 / 3.2 us for K's one, profiles/r4_run_h_prepass_trace.txt)
                    if constexpr (IS_V) va[u] = __hip_atomic_load(wi + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if constexpr (IS_V && !V_AMAX) vc[u] = __hip_atomic_load(wi + D + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if constexpr (!V_AMAX) vs[u] = __hip_atomic_load(wi + 2 * D + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    if (i0 + u < nslab) {
                        if constexpr (IS_V) a = fmaxf(a, va[u]);
                        if constexpr (IS_V && !V_AMAX) c = fminf(c, vc[u]);
                        if constexpr (!V_AMAX) s += vs[u];
                    }
                }
            }
        }
        if (tid < D) {
            if constexpr (!IS_V) {
                const uint16_t m = st16<DT>(s / (float)(VARLEN ? p.L : L));      // k.mean(dim=seq) in the input dtype, one rounding (VARLEN: over all packed tokens)
                kmean[tid] = m;
                if (gslab == 0) reinterpret_cast<uint16_t *>(p.k_mean)[bh * D + tid] = m;
            } else {
                // per-channel scale (and mean for smooth_v): the rules of prep_v_kernel (fused.cu:335-395)
                const bool smooth = p.v_mean != nullptr;
                const float lpad = (float)((L + 15) / 16 * 16);
                const float mean = smooth ? s / lpad : 0.0f;
                if constexpr (V_AMAX) c = 0.0f;                      // (max, min) = (abs-max, 0)
                if ((L & 15) != 0) { a = fmaxf(a, 0.0f); c = fminf(c, 0.0f); }
                const float am = fmaxf(fabsf(a - mean), fabsf(c - mean));
                ch_mean[tid] = mean;
                ch_recp[tid] = am > 0.0f ? p.scale_max / am : 0.0f;
                if (gslab == 0) {
                    p.v_scale[bh * D + tid] = am / p.scale_max;
                    if (smooth) p.v_mean[bh * D + tid] = mean;
                }
            }
        }
        __syncthreads();
        SAGE_STAMP();                              // 4: head statistics reduced
        if (nslab > 1 && tid == 0 && !(SAGE_PP_ABL & 1)) {          // the last slab to leave re-arms the head's counters
            const unsigned left = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (left == (unsigned)nslab - 1) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (gap) return;                                        // (workgroup-uniform; the slab has arrived, departed and is needed no further)
    if (SAGE_PP_ABL & 2) {
        unsigned acc = 0;
#pragma unroll
        for (int i = 0; i < NR; i++) acc += rw[i][0] + rw[i][1];
        if (acc == 0x12345u) p.ws[0] = 1.0f;
        return;
    }

    if constexpr (!IS_V) {
        // ---- 3a. K: INT8 rows + group scales (the arithmetic of quant_int8_kernel, on the registers) --------------------
        const int blk = p.k_blk;                            // 64 / 128 keys per scale block (one group map per block)
        const int bsh = (blk == 128) ? 7 : 6;
        const int nb = kStatsSlab >> bsh;                   // blocks per slab
        const int ngroups = (p.k_gran == GR_BLOCK) ? 1 : 4 * (blk / p.k_warp);
        const bool smooth = p.k_mean != nullptr;
        const unsigned init_bits = (p.k_style == QS_CUDA) ? __float_as_uint(1e-7f) : 0u;    // fused.cu:147
        if (tid < 64) gmax[tid >> 3][tid & 7] = init_bits;
        float mean4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        unsigned m01 = 0u, m23 = 0u;
        if (smooth) {
            m01 = (unsigned)kmean[c4] | ((unsigned)kmean[c4 + 1] << 16);
            m23 = (unsigned)kmean[c4 + 2] | ((unsigned)kmean[c4 + 3] << 16);
#pragma unroll
            for (int j = 0; j < 4; j++) mean4[j] = ld16<DT>(kmean[c4 + j]);
        }
        // rows past the end (zeros from the buffer loads) take the mean's bits: their smoothed value is exactly 0, so the two
        // passes below need no per-row validity test.  In place and without selects: OR with the mean under a sign mask.
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const unsigned gone = (unsigned)((my_rows - 1 - i) >> 31);       // all ones when row i does not exist
            rw[i][0] |= m01 & gone;
            rw[i][1] |= m23 & gone;
        }
#if SAGE_PP_KWB
        // Triton rounding styles: `k - km` is rounded to the input dtype before anything else looks at it (core.py:281,
        // quant_per_block.py:53-54), so the smoothed, rounded pair REPLACES the raw word once, here, and the two passes below run their
        // un-smoothed forms on it (forming the pair in each pass cost 2.5 VALU instructions per element more; doing it inside the
        // abs-max pass tipped the register allocation into 20 spilled VGPRs)
        if (smooth && p.k_style != QS_CUDA) {
#pragma unroll
            for (int i = 0; i < NR; i++) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    unsigned u = rw[i][c];
                    const float lo = ld16<DT>((uint16_t)(u & 0xffffu)) - mean4[2 * c];
                    const float hi = ld16<DT>((uint16_t)(u >> 16)) - mean4[2 * c + 1];
                    u = pack_pair_to_dtype<DT>(lo, hi);
                    asm volatile("" : "+v"(u));
                    rw[i][c] = u;
                }
                asm volatile("" ::: "memory");
            }
        }
#endif
        __syncthreads();
        const int g_thread = group_of_row(r0 & (blk - 1), p.k_gran, p.k_warp);   // every row of a thread in a block: same group
        const int nblk_total = (L + blk - 1) >> bsh;
        // buffer stores: rows past the end of the head are dropped by the range check
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.k_out + (long)b * p.ko_sb + (long)h * p.ko_sh + tok0 * p.ko_sl, 0, (unsigned)((long)(L - 1) * p.ko_sl + D), 0x00020000);
        const unsigned ooff = (unsigned)((row0 + r0) * (int)p.ko_sl + c4), ostep = (unsigned)(RPI * (int)p.ko_sl);

        // STYLE: 0 the CUDA quantiser (fp32 difference, round-to-nearest-even, fused.cu:131-172), 1 Triton rounding with
        // the epsilon scale (quant_per_thread.py:41-44) -- the two K conventions of the reference's CUDA entry points --,
        // 2 Triton rounding with the plain scale amax / 127, zero for an all-zero group (quant_per_block.py:41-44: the Triton-named API)
        auto quantise = [&](auto style_tag, auto smooth_tag) {
            constexpr int STYLE = decltype(style_tag)::value;
            constexpr bool SMOOTH = decltype(smooth_tag)::value;
            // the 4 values of row i: (k - km), rounded to the input dtype unless the CUDA quantiser's fp32 difference is
            // asked for (quant_per_block.py:53-54 vs fused.cu:131-137); pre_scale is 1 for K
            auto row_values = [&](int i, float (&f)[4]) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    unsigned w = rw[i][c];
                    // every pass recomputes the values from the packed word; without this the compiler recognises the
                    // common subexpressions and carries 128 unpacked floats from pass to pass (1096 VGPRs spilled)
                    asm volatile("" : "+v"(w));
                    float lo = ld16<DT>((uint16_t)(w & 0xffffu)), hi = ld16<DT>((uint16_t)(w >> 16));
                    if constexpr (SMOOTH) {
                        lo -= mean4[2 * c];
                        hi -= mean4[2 * c + 1];
                        if (STYLE != 0) round_pair_to_dtype<DT>(lo, hi);
                    }
                    f[2 * c] = lo;
                    f[2 * c + 1] = hi;
                }
            };
            // A thread's rows inside one scale block (rb of them, consecutive i) share their group, and so do the 32 lanes
            // around it: rows r0 and r0 ^ 1 map to the same group in every granularity.  Accumulate per block, reduce over
            // the half-wave with DPP, one LDS atomic per half-wave and block (one per thread and row pair serialised the LDS
            // unit: 32 lanes per address, 16 us per workgroup in this pass alone).
            const int rb_mask = (blk / RPI) - 1;                 // rows of a thread per block, minus 1 (1, 3 or 7)
            float amax = 0.0f;
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            us2 amax16 = {0, 0};
#pragma unroll
            for (int i = 0; i < NR; i++) {
                if constexpr (STYLE == 0) {
                    float f[4];
                    row_values(i, f);
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(f[0]), fabsf(f[1]))), fmaxf(fabsf(f[2]), fabsf(f[3])));
                } else {
                    // values rounded to the input dtype (quant_per_block.py:53-54): the abs-max runs on the 16-bit patterns, which
                    // order like unsigned integers in a sign-magnitude format (SAGE_PP_KWB: rw already holds the smoothed, rounded
                    // pairs and SMOOTH is false here)
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        unsigned u = rw[i][c];
                        if constexpr (SMOOTH) {
                            asm volatile("" : "+v"(u));
                            const float lo = ld16<DT>((uint16_t)(u & 0xffffu)) - mean4[2 * c];
                            const float hi = ld16<DT>((uint16_t)(u >> 16)) - mean4[2 * c + 1];
                            u = pack_pair_to_dtype<DT>(lo, hi);
                        }
                        const unsigned mag = u & 0x7fff7fffu;
                        us2 m2;
                        __builtin_memcpy(&m2, &mag, 4);
                        amax16 = __builtin_elementwise_max(amax16, m2);
                    }
                }
                if ((i & 1) && ((i & rb_mask) == rb_mask)) {     // last row of the block (uniform; blocks hold >= 2 rows)
                    float m = amax;
                    if constexpr (STYLE != 0) { m = ld16<DT>(amax16[0] > amax16[1] ? amax16[0] : amax16[1]); amax16 = us2{0, 0}; }
                    m = fmaxf(m, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(m), 0xB1, 0xf, 0xf, true)));    // lane ^ 1
                    m = fmaxf(m, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(m), 0x4E, 0xf, 0xf, true)));    // lane ^ 2
                    m = fmaxf(m, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(m), 0x141, 0xf, 0xf, true)));   // 7 - lane (of 8)
                    m = fmaxf(m, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(m), 0x140, 0xf, 0xf, true)));   // 15 - lane (of 16)
                    m = fmaxf(m, __shfl_xor(m, 16));
                    if ((tid & 31) == 0) atomicMax(&gmax[(i * RPI) >> bsh][g_thread], __float_as_uint(m));
                    amax = 0.0f;
                }
                asm volatile("" : "+v"(amax), "+v"(amax16) :: "memory");      // keep the rows in order (hoisting them all spills): tie the accumulators
            }
            __syncthreads();
            SAGE_STAMP();                          // 5 (K): group maxima
            if (tid < nb * ngroups) {
                const int kb = tid / ngroups, g = tid % ngroups;
                const int gb = slab * nb + kb;
                const float am = __uint_as_float(gmax[kb][g]);
                const float sc = quant_scale(am, p.k_style);
                gsc[kb][g] = sc;
                gy[kb][g] = (STYLE == 0) ? 127.0f / am : quant_recip(sc);       // fused.cu:164
                float *ksc_out = VARLEN ? p.k_scale + (tile0 + gb) * p.H + h                   // [sum nblk, H] (quant_per_block_varlen.py:75-76)
                                        : p.k_scale + (bh * nblk_total + gb) * ngroups + g;
                if (gb < nblk_total) *ksc_out = sc;
            }
            __syncthreads();
            unsigned orun = ooff;
#pragma unroll
            for (int i = 0; i < NR; i += 2) {
                const float sc = gsc[(i * RPI) >> bsh][g_thread], y = gy[(i * RPI) >> bsh][g_thread];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    float f[4];
                    row_values(i + u, f);
                    int q[4];
                    if constexpr (STYLE == 0) {
#pragma unroll
                        for (int j = 0; j < 4; j++) q[j] = quant_round_cuda(f[j], y);
                    } else if constexpr (STYLE == 1) quant_round_triton_nz4(f, sc, y, q);      // (the 4-wide forms of sage_quant_math.h: same bits)
                    else quant_round_triton4(f, sc, y, q);
                    __builtin_amdgcn_raw_buffer_store_b32(pack_int8x4(q[0], q[1], q[2], q[3]), orsrc, orun, 0, SAGE_PP_NT ? 2 : 0);
                    orun += ostep;
                    asm volatile("" : "+v"(orun) :: "memory");
                }
            }
        };
        // one instantiation per (rounding style, smooth_k?): no run-time branch inside the passes
        auto by_smooth = [&](auto style_tag) {
            if (smooth) quantise(style_tag, std::true_type{}); else quantise(style_tag, std::false_type{});
        };
        if (p.k_style == QS_CUDA) by_smooth(std::integral_constant<int, 0>{});
#if SAGE_PP_KWB
        else if (p.k_style == QS_TRITON_THREAD) quantise(std::integral_constant<int, 1>{}, std::false_type{});
        else quantise(std::integral_constant<int, 2>{}, std::false_type{});          // QS_TRITON (the C ABI admits these three)
#else
        else if (p.k_style == QS_TRITON_THREAD) by_smooth(std::integral_constant<int, 1>{});
        else by_smooth(std::integral_constant<int, 2>{});          // QS_TRITON (the C ABI admits these three)
#endif
    } else {
        // ---- 3b. V: tile image, two 64-token tiles per LDS stage (the arithmetic of prep_v_kernel) ------------------------
        const bool smooth = V_AMAX ? false : (p.v_mean != nullptr);      // (V_AMAX: known at compile time -- as a run-time flag the subtraction,
                                                                             //  the row test and two v_cndmask per element stayed in the loop)
        const int ntiles = (L + BLKK - 1) / BLKK;
        constexpr int RPS = 2 * BLKK / RPI;                 // rows of a thread per stage
        auto stage_in = [&](int s) {                        // the stage's 128 rows -> LDS
#pragma unroll
            for (int m = 0; m < RPS; m++) {
                const int rs = r0 + m * RPI;                // row inside the stage (0..127)
                const v2u t = {rw[s * RPS + m][0], rw[s * RPS + m][1]};
                *reinterpret_cast<v2u *>(&tile[rs >> 6][(rs & 63) * LDT + c4]) = t;
            }
        };
        if (p.v_fp16 == 0) {
            unsigned char *img = reinterpret_cast<unsigned char *>(p.v_image) + bh * (long)ntiles * (D * 64);      // (dense only: the varlen call has no FP8 image)
#pragma unroll
            for (int s = 0; s < kStatsSlab / (2 * BLKK); s++) {
                if (row0 + s * 2 * BLKK < L) {                  // workgroup-uniform
                    stage_in(s);
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < 2 * D * 4 / NT; it++) {
                        const int piece = tid + NT * it;
                        const int tt = piece / (D * 4), pin = piece % (D * 4);
                        const int d = pin >> 2, pc = pin & 3;
                        const int t = (row0 >> 6) + 2 * s + tt;
                        if (t >= ntiles) continue;
                        const int ch = swz_chunk<64>(d, pc);
                        const float mean = ch_mean[d], recp = ch_recp[d];
                        float f[16];
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const int tok = pv_token_of_position(16 * ch + j);
                            float xv = ld16<DT>(tile[tt][tok * LDT + d]);
                            if (smooth) xv = (t * BLKK + tok < L) ? xv - mean : 0.0f;      // padding stays zero
                            xv *= recp;
                            f[j] = fminf(fmaxf(xv, -448.0f), 448.0f);                      // satfinite
                        }
                        v4u pk;
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            int word = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * w], f[4 * w + 1], 0, false);
                            word = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * w + 2], f[4 * w + 3], word, true);
                            pk[w] = (unsigned)word;
                        }
                        if constexpr (SAGE_PP_NT != 0) __builtin_nontemporal_store(pk, reinterpret_cast<v4u *>(img + (long)t * (D * 64) + d * 64 + pc * 16));
                        else *reinterpret_cast<v4u *>(img + (long)t * (D * 64) + d * 64 + pc * 16) = pk;
                    }
                    __syncthreads();
                }
            }
        } else {
            // FP16-PV entry points: `v.to(float16)` (core.py:297-298,613) as the fp16 tile image of prep_v_kernel -- no
            // statistics, no barrier: load, transpose through LDS, store
            // dense: [B, H, ntiles, D, 64]; VARLEN: [sum ntiles, H, D, 64] (tile-major, the layout of prep_v_kernel's packed form)
            const long img_t = VARLEN ? (long)p.H * (D * 128) : (long)(D * 128);
            unsigned char *img = reinterpret_cast<unsigned char *>(p.v_image) + (VARLEN ? (tile0 * p.H + h) * (long)(D * 128) : bh * (long)ntiles * (D * 128));
#pragma unroll
            for (int s = 0; s < kStatsSlab / (2 * BLKK); s++) {
                if (row0 + s * 2 * BLKK < L) {
                    stage_in(s);
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < 2 * D * 8 / NT; it++) {
                        const int piece = tid + NT * it;
                        const int tt = piece / (D * 8), pin = piece % (D * 8);
                        const int d = pin >> 3, pc = pin & 7;
                        const int t = (row0 >> 6) + 2 * s + tt;
                        if (t >= ntiles) continue;
                        const int ch = swz_chunk<128>(d, pc);
                        v4u pk;
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            unsigned word = 0;
#pragma unroll
                            for (int e = 0; e < 2; e++) {
                                const int tok = pv_token_of_position(8 * ch + 2 * w + e);
                                uint16_t raw = tile[tt][tok * LDT + d];
                                if (DT != DT_F16) raw = f32_to_f16_rne(bf16_to_f32(raw));
                                word |= (unsigned)raw << (16 * e);
                            }
                            pk[w] = word;
                        }
                        if constexpr (SAGE_PP_NT != 0) __builtin_nontemporal_store(pk, reinterpret_cast<v4u *>(img + (long)t * img_t + d * 128 + pc * 16));
                        else *reinterpret_cast<v4u *>(img + (long)t * img_t + d * 128 + pc * 16) = pk;
                    }
                    __syncthreads();
                }
            }
        }
    }
#if SAGE_PP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (is_v) tslot = 6;
    SAGE_STAMP();                                  // 6: stores acknowledged
    if (tid == 0) trace[7] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) |              // XCC_ID
                              ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8);     // HW_ID
#endif
}

template <int D, int DT, bool VARLEN>
#ifndef SAGE_PP_WAVES      // waves per SIMD the allocator must allow (4 = two 512-thread workgroups per CU)
#define SAGE_PP_WAVES 4
#endif
#ifndef SAGE_PP_WAVES64    // D = 64 holds half the data registers: three workgroups per CU (<= 85 VGPRs).  Left to the default bound the bf16
#define SAGE_PP_WAVES64 6  // instantiation drifted from 80 to 93 VGPRs with a source change and lost the third workgroup: 204 -> 235 us at C5
#endif
__global__ void __launch_bounds__(kPrepassThreads, (D == 64 ? SAGE_PP_WAVES64 : SAGE_PP_WAVES))
prepass_kv_kernel(const PrepassParams p)
{
    __shared__ PrepassLds<D> lds;
    // dispatch order: every head of K of one batch element, then every head of its V (alternating K and V heads instead
    // measured the same at C3 and 15 % slower at B=16 H=32 N=1024)
    const int is_v = (p.parts == 3) ? (int)(blockIdx.z & 1) : (p.parts == 2);
    const int b = (p.parts == 3) ? (int)(blockIdx.z >> 1) : (int)blockIdx.z;
    if (is_v) {
        if (SAGE_PP_VAMAX && !VARLEN && p.v_mean == nullptr && p.v_fp16 == 0) prepass_body<D, DT, true, VARLEN, true>(p, lds, b);
        else prepass_body<D, DT, true, VARLEN>(p, lds, b);
    } else prepass_body<D, DT, false, VARLEN>(p, lds, b);
}

__global__ void __launch_bounds__(1024) debug_spin_kernel(long long ticks)
{
    const long long t0 = wall_clock64();                // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

hipError_t launch_debug_spin(int ms, int nwg, hipStream_t s)
{
    hipLaunchKernelGGL(debug_spin_kernel, dim3(nwg), dim3(1024), 0, s, (long long)ms * 100000LL);
    return hipGetLastError();
}

hipError_t launch_prepass_kv(const PrepassParams &p, hipStream_t s)
{
    if (p.B <= 0 || p.H <= 0 || p.nslab <= 0 || p.parts == 0) return hipSuccess;
    dim3 grid(p.nslab, p.H, p.B * (p.parts == 3 ? 2 : 1));
#define SAGE_PP(D_, T_) do { if (p.cu != nullptr) hipLaunchKernelGGL((prepass_kv_kernel<D_, T_, true>), grid, dim3(kPrepassThreads), 0, s, p); \
                             else hipLaunchKernelGGL((prepass_kv_kernel<D_, T_, false>), grid, dim3(kPrepassThreads), 0, s, p); } while (0)
    if (p.D == 128) { if (p.dtype == DT_F16) SAGE_PP(128, DT_F16); else SAGE_PP(128, DT_BF16); }
    else if (p.D == 64) { if (p.dtype == DT_F16) SAGE_PP(64, DT_F16); else SAGE_PP(64, DT_BF16); }
    else return hipErrorInvalidValue;
#undef SAGE_PP
    return hipGetLastError();
}

}  // namespace sage
