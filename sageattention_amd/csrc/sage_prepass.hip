// sage_prepass.hip -- the K / V pre-pass of one attention call as ONE launch that reads K and V once.
//
// What it replaces (same bits, see tests/test_gpu_prepass.py):
//   K: stats_partial + stats_final (k.mean(dim=seq), core.py:280) + quant_int8_kernel (fused.cu:64-198 /
//      quant_per_thread.py:58-98 with the `k - km` of core.py:281 fused)                       3 launches, 4 B/elt read
//   V: stats_partial + stats_final + prep_v_kernel (fused.cu:262-427, quant.py:224-293)         3 launches, 4 B/elt read
// A workgroup owns one 512-token slab of one (batch, kv-head) of K or of V and keeps it in REGISTERS (32 x 16 B per
// thread at D = 128) across the three steps
//   1. per-channel (max, min, sum) of the slab -> workspace          (same summation order as sage_stats.hip)
//   2. wait until the other slabs of the head have published theirs; reduce them in slab order
//   3. quantise the slab from the registers: INT8 rows + group scales (K), FP8 PV-operand tile image (V)
// so HBM sees 2 B/elt in and 1 B/elt out -- the algorithmic minimum -- instead of 4 + 1.
//
// Step 2 is a per-head barrier between workgroups of one launch.  It is safe because (a) gfx950 hands workgroups to
// its 8 XCDs round-robin and each XCD starts its share in index order, (b) the slabs of a head are consecutive in
// index, so the lowest unfinished head always has every slab resident or next in line, and (c) the C ABI refuses
// heads of more than kPrepassMaxSlabs slabs (8 per XCD), far below the resident-workgroup capacity (2 per CU).
// The same assumption carries rocPRIM's decoupled look-back scan.  Cross-XCD visibility of the partials follows the
// gfx942+ memory model for atomics: the arrival counter and the partials are agent-scope atomic accesses.
// The two counters of a head return to zero before the kernel ends, so one zero-initialised sync buffer serves every
// call issued in stream order.
#include "sage_common.h"
#include "sage_kernels.h"
#include "sage_quant_math.h"
#include <type_traits>

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

template <int D, int DT>
__global__ void __launch_bounds__(kPrepassThreads, 4)
prepass_kv_kernel(const PrepassParams p)
{
    // 512 threads, 4 channels (8 B) per thread and row: a row is D / 4 threads wide, the workgroup passes over RPI rows at
    // a time and thread (r0, c4) owns rows r0 + RPI * i -- the row -> thread map of stats_partial_kernel (RPI = 16 at D = 128,
    // 32 at D = 64), so the per-channel sums associate the same way.  64 data VGPRs per thread (D = 128) leave room for
    // 4 waves / SIMD: two slabs per CU as with 256 threads x 8 channels, but twice the waves to hide the latency of the
    // compute steps (which, not HBM, bounded the 256-thread version: 194 us against 154 us for the six launches at C3).
    constexpr int NT = kPrepassThreads;
    constexpr int TPR = D / 4;                  // threads per row
    constexpr int RPI = NT / TPR;               // rows per pass of the workgroup
    constexpr int NR = kStatsSlab / RPI;        // rows per thread: 32 (D = 128) / 16 (D = 64)
    constexpr int LDT = D + 8;
    // The slab lives in two register tuples (dword c of row i = rw[c][i]) that the loops below index with a wave-uniform
    // counter (s_set_gpr_idx): rolled loops keep the kernel at ~2k instructions.  Fully unrolled it was 20k instructions
    // (120 KB) of straight-line code that every wave fetched exactly once -- instruction fetch set its speed.
    typedef unsigned rows_t __attribute__((ext_vector_type(NR)));
    __shared__ float red[3][RPI][D];
    __shared__ __attribute__((aligned(16))) uint16_t tile[2][BLKK * LDT];
    __shared__ float ch_mean[D], ch_recp[D];
    __shared__ uint16_t kmean[D];
    __shared__ unsigned gmax[8][8];

    const int tid = threadIdx.x;
    const int slab = blockIdx.x, h = blockIdx.y;
    const int is_v = (p.parts == 3) ? (int)(blockIdx.z & 1) : (p.parts == 2);
    const int b = (p.parts == 3) ? (int)(blockIdx.z >> 1) : (int)blockIdx.z;
    const int L = p.L;
    const long bh = (long)b * p.H + h;
    const uint16_t *x = is_v ? reinterpret_cast<const uint16_t *>(p.v) + (long)b * p.v_sb + (long)h * p.v_sh
                             : reinterpret_cast<const uint16_t *>(p.k) + (long)b * p.k_sb + (long)h * p.k_sh;
    const long x_sl = is_v ? p.v_sl : p.k_sl;
    const int c4 = (tid % TPR) * 4, r0 = tid / TPR;
    const int row0 = slab * kStatsSlab;
    const int end = min(L, row0 + kStatsSlab);
    const int my_rows = (end - row0 - r0 + RPI - 1) / RPI;      // rows i < my_rows of this thread exist (may be <= 0)

    // ---- the slab, one read -------------------------------------------------------------------------------------
    rows_t rw[2];
#pragma unroll
    for (int i = 0; i < NR; i++) {
        v2u t = {0u, 0u};
        if (i < my_rows) t = *reinterpret_cast<const v2u *>(x + (long)(row0 + r0 + i * RPI) * x_sl + c4);
        rw[0][i] = t[0]; rw[1][i] = t[1];
    }

    const bool need_stats = is_v || p.k_mean != nullptr;
    if (need_stats) {
        // ---- 1. slab statistics, rows in index order (bit-compatible with stats_partial_kernel) ----------------------
        float mx[4], mn[4], sm[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { mx[j] = -INFINITY; mn[j] = INFINITY; sm[j] = 0.0f; }
        if (!(SAGE_PP_ABL & 4)) {
#pragma unroll 1
            for (int g = 0; g < NR; g += 4) {
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int i = g + m;
                    if (i < my_rows) {
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            const unsigned w = rw[c][i];
                            const float lo = ld16<DT>((uint16_t)(w & 0xffffu)), hi = ld16<DT>((uint16_t)(w >> 16));
                            mx[2 * c] = fmaxf(mx[2 * c], lo);         mn[2 * c] = fminf(mn[2 * c], lo);         sm[2 * c] += lo;
                            mx[2 * c + 1] = fmaxf(mx[2 * c + 1], hi); mn[2 * c + 1] = fminf(mn[2 * c + 1], hi); sm[2 * c + 1] += hi;
                        }
                    }
                }
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
            float *mine = ws + (long)slab * 3 * D;
            __hip_atomic_store(mine + tid, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + D + tid, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 2 * D + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- 2. per-head barrier over the slabs, then the reduction in slab order (as stats_final_kernel) -------------
        unsigned *cnt = p.sync + ((long)is_v * p.B * p.H + bh) * kPrepassSyncStride;   // one 128-B line per head
        if (p.nslab > 1 && !(SAGE_PP_ABL & 1)) {
            // Everything that crosses workgroups here is an agent-scope atomic access (write-through / cache-bypassing,
            // coherent across the XCDs by itself), so the ordering needs no L2 write-back or invalidate -- a fence at
            // agent scope would put `buffer_wbl2` / `buffer_inv` into every spin iteration of every waiting workgroup
            // (measured: 2.3x slower).  Partials first, then the arrival: the stores are acknowledged (vmcnt) before
            // thread 0 counts the slab in; the readers issue their loads after the spin.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.nslab)
                    __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        }
        if (tid < D) {
            // slabs in index order; eight slabs' loads are in flight together (one round trip to the coherence point per batch,
            // not one per slab: with the loads issued one by one this loop alone cost ~2 us x nslab per workgroup)
            float a = -INFINITY, c = INFINITY, s = 0.0f;
            for (int i0 = 0; i0 < p.nslab; i0 += 8) {
                float va[8], vc[8], vs[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float *wi = ws + (long)min(i0 + u, p.nslab - 1) * 3 * D;
                    va[u] = __hip_atomic_load(wi + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vc[u] = __hip_atomic_load(wi + D + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vs[u] = __hip_atomic_load(wi + 2 * D + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (i0 + u < p.nslab) { a = fmaxf(a, va[u]); c = fminf(c, vc[u]); s += vs[u]; }
                }
            }
            if (!is_v) {
                const uint16_t m = st16<DT>(s / (float)L);           // k.mean(dim=seq) in the input dtype, one rounding
                kmean[tid] = m;
                if (slab == 0) reinterpret_cast<uint16_t *>(p.k_mean)[bh * D + tid] = m;
            } else {
                // per-channel scale (and mean for smooth_v): the rules of prep_v_kernel (fused.cu:335-395)
                const bool smooth = p.v_mean != nullptr;
                const float lpad = (float)((L + 15) / 16 * 16);
                const float mean = smooth ? s / lpad : 0.0f;
                if ((L & 15) != 0) { a = fmaxf(a, 0.0f); c = fminf(c, 0.0f); }
                const float am = fmaxf(fabsf(a - mean), fabsf(c - mean));
                ch_mean[tid] = mean;
                ch_recp[tid] = am > 0.0f ? p.scale_max / am : 0.0f;
                if (slab == 0) {
                    p.v_scale[bh * D + tid] = am / p.scale_max;
                    if (smooth) p.v_mean[bh * D + tid] = mean;
                }
            }
        }
        __syncthreads();
        if (p.nslab > 1 && tid == 0 && !(SAGE_PP_ABL & 1)) {          // the last slab to leave re-arms the head's counters
            const unsigned left = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (left == (unsigned)p.nslab - 1) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (SAGE_PP_ABL & 2) {
        unsigned acc = 0;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < NR; i++) acc += rw[c][i];
        if (acc == 0x12345u) p.ws[0] = 1.0f;
        return;
    }

    if (!is_v) {
        // ---- 3a. K: INT8 rows + group scales (the arithmetic of quant_int8_kernel, on the registers) --------------------
        const int blk = p.k_blk;                            // 64 / 128 keys per scale block (one group map per block)
        const int bsh = (blk == 128) ? 7 : 6;
        const int nb = kStatsSlab >> bsh;                   // blocks per slab
        const int rb = blk / RPI;                           // rows of a thread inside one block (consecutive i): 2, 4 or 8
        const int ngroups = (p.k_gran == GR_BLOCK) ? 1 : 4 * (blk / p.k_warp);
        const bool smooth = p.k_mean != nullptr;
        const unsigned init_bits = (p.k_style == QS_CUDA) ? __float_as_uint(1e-7f) : 0u;    // fused.cu:147
        if (tid < 64) gmax[tid >> 3][tid & 7] = init_bits;
        float mean4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) mean4[j] = smooth ? ld16<DT>(kmean[c4 + j]) : 0.0f;
        __syncthreads();
        const int g_thread = group_of_row(r0 & (blk - 1), p.k_gran, p.k_warp);   // every row of a thread in a block: same group
        const int nblk_total = (L + blk - 1) >> bsh;
        int8_t *out = p.k_out + (long)b * p.ko_sb + (long)h * p.ko_sh + (long)(row0 + r0) * p.ko_sl + c4;

        // STYLE: 0 the CUDA quantiser (fp32 difference, round-to-nearest-even, fused.cu:131-172), 1 Triton rounding with
        // the epsilon scale (quant_per_thread.py:41-44), 2 Triton rounding, plain scale (quant_per_block.py:41-44)
        auto quantise = [&](auto style_tag) {
            constexpr int STYLE = decltype(style_tag)::value;
            // the 4 values of row i: (k - km), rounded to the input dtype unless the CUDA quantiser's fp32 difference is
            // asked for (quant_per_block.py:53-54 vs fused.cu:131-137); pre_scale is 1 for K
            auto row_values = [&](int i, float (&f)[4]) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const unsigned w = rw[c][i];
                    float lo = ld16<DT>((uint16_t)(w & 0xffffu)), hi = ld16<DT>((uint16_t)(w >> 16));
                    if (smooth) {
                        lo -= mean4[2 * c];
                        hi -= mean4[2 * c + 1];
                        if (STYLE != 0) round_pair_to_dtype<DT>(lo, hi);
                    }
                    f[2 * c] = lo;
                    f[2 * c + 1] = hi;
                }
            };
#pragma unroll 1
            for (int kb = 0; kb < nb; kb++) {
                float amax = 0.0f;
#pragma unroll 1
                for (int m = 0; m < rb; m += 2) {
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int i = kb * rb + m + u;
                        if (i < my_rows) {
                            float f[4];
                            row_values(i, f);
#pragma unroll
                            for (int j = 0; j < 4; j++) amax = fmaxf(amax, fabsf(f[j]));
                        }
                    }
                }
                if (kb * rb < my_rows) atomicMax(&gmax[kb][g_thread], __float_as_uint(amax));
            }
            __syncthreads();
            if (tid < nb * ngroups) {
                const int kb = tid / ngroups, g = tid % ngroups;
                const int gb = slab * nb + kb;
                if (gb < nblk_total)
                    p.k_scale[(bh * nblk_total + gb) * ngroups + g] = quant_scale(__uint_as_float(gmax[kb][g]), p.k_style);
            }
#pragma unroll 1
            for (int kb = 0; kb < nb; kb++) {
                const float am = __uint_as_float(gmax[kb][g_thread]);
                const float inv = 127.0f / am;                               // fused.cu:164
                const float sc = quant_scale(am, p.k_style);
                const float y = quant_recip(sc);
#pragma unroll 1
                for (int m = 0; m < rb; m += 2) {
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int i = kb * rb + m + u;
                        if (i >= my_rows) continue;
                        float f[4];
                        row_values(i, f);
                        int q[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if constexpr (STYLE == 0) q[j] = quant_round_cuda(f[j], inv);
                            else if constexpr (STYLE == 1) q[j] = quant_round_triton_nz(f[j], sc, y);
                            else q[j] = quant_round_triton(f[j], sc, y);
                        }
                        *reinterpret_cast<unsigned *>(out + (long)(i * RPI) * p.ko_sl) = pack_int8x4(q[0], q[1], q[2], q[3]);
                    }
                }
            }
        };
        if (p.k_style == QS_CUDA) quantise(std::integral_constant<int, 0>{});
        else if (p.k_style == QS_TRITON_THREAD) quantise(std::integral_constant<int, 1>{});
        else quantise(std::integral_constant<int, 2>{});
    } else {
        // ---- 3b. V: FP8 tile image, two 64-token tiles per LDS stage (the arithmetic of prep_v_kernel) -------------------
        const bool smooth = p.v_mean != nullptr;
        const int ntiles = (L + BLKK - 1) / BLKK;
        unsigned char *img = reinterpret_cast<unsigned char *>(p.v_image) + bh * (long)ntiles * (D * 64);
        constexpr int RPS = 2 * BLKK / RPI;                 // rows of a thread per stage
#pragma unroll 1
        for (int s = 0; s < kStatsSlab / (2 * BLKK); s++) {
            if (row0 + s * 2 * BLKK >= L) break;            // workgroup-uniform
#pragma unroll
            for (int m = 0; m < RPS; m++) {
                const int rs = r0 + m * RPI;                // row inside the stage (0..127)
                const int i = s * RPS + m;
                const v2u t = {rw[0][i], rw[1][i]};
                *reinterpret_cast<v2u *>(&tile[rs >> 6][(rs & 63) * LDT + c4]) = t;
            }
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
                *reinterpret_cast<v4u *>(img + (long)t * (D * 64) + d * 64 + pc * 16) = pk;
            }
            __syncthreads();
        }
    }
}

hipError_t launch_prepass_kv(const PrepassParams &p, hipStream_t s)
{
    if (p.B <= 0 || p.H <= 0 || p.nslab <= 0 || p.parts == 0) return hipSuccess;
    dim3 grid(p.nslab, p.H, p.B * (p.parts == 3 ? 2 : 1));
#define SAGE_PP(D_, T_) hipLaunchKernelGGL((prepass_kv_kernel<D_, T_>), grid, dim3(kPrepassThreads), 0, s, p)
    if (p.D == 128) { if (p.dtype == DT_F16) SAGE_PP(128, DT_F16); else SAGE_PP(128, DT_BF16); }
    else if (p.D == 64) { if (p.dtype == DT_F16) SAGE_PP(64, DT_F16); else SAGE_PP(64, DT_BF16); }
    else return hipErrorInvalidValue;
#undef SAGE_PP
    return hipGetLastError();
}

}  // namespace sage
