// sage_quant.hip -- INT8 quantisation of Q / K with fused K-mean subtraction (HBM-bound).
//
// One kernel covers the reference's four quantisers:
//   sageattention/triton/quant_per_block.py:21-47        per-block, "triton" rounding
//   sageattention/triton/quant_per_block_varlen.py:21-58 same per packed sequence
//   csrc/fused/fused.cu:64-198 (QuantInt8Kernel)          per-block / per-warp, "cuda" rounding,
//                                                         fused `- mean`, fused `* sm_scale`
//   sageattention/triton/quant_per_thread.py:21-98        "per-thread" row groups
// A workgroup owns BLK consecutive rows of one (batch, head); every thread keeps its 16-element
// chunks in registers between the abs-max pass and the rounding pass, so the tensor is read
// exactly once (2 B/elt in, 1 B/elt out -- the algorithmic minimum).  Group maxima are combined
// with LDS atomicMax on the IEEE bit pattern (values are non-negative).
#include "sage_common.h"
#include "sage_kernels.h"
#include "sage_quant_math.h"

namespace sage {

template <int D, int BLK, int DT>
__global__ void __launch_bounds__(256)
quant_int8_kernel(const QuantParams p)
{
    constexpr int CPR = D / 16;                 // 16-element chunks per row
    constexpr int NCH = BLK * CPR / 256;        // chunks per thread
    __shared__ unsigned gmax[64];          // up to (128 / 16) * 8 per-thread groups

    const int tid = threadIdx.x;
    const int blk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    int L = p.L;
    long xoff, ooff;
    float *sc_out;
    int ngroups;
    if (p.gran == GR_BLOCK) ngroups = 1;
    else if (p.gran == GR_WARP) ngroups = BLK / p.warp;
    else if (p.gran == GR_THREAD_Q) ngroups = (BLK / p.warp) * 8;
    else ngroups = (BLK / p.warp) * 4;

    if (p.cu != nullptr) {
        const int s0 = p.cu[b];
        L = p.cu[b + 1] - s0;
        if (blk * BLK >= L) return;
        xoff = (long)s0 * p.x_sl + (long)h * p.x_sh;
        ooff = (long)s0 * p.o_sl + (long)h * p.o_sh;
        sc_out = p.scale + ((long)p.cu_scale[b] + blk) * p.H + h;       // [sum nblk, H]
    } else {
        xoff = (long)b * p.x_sb + (long)h * p.x_sh;
        ooff = (long)b * p.o_sb + (long)h * p.o_sh;
        sc_out = p.scale + ((long)b * p.H + h) * p.nscale + (long)blk * ngroups;
    }
    const unsigned init_bits = (p.style == QS_CUDA) ? __float_as_uint(1e-7f) : 0u;   // fused.cu:147
    if (tid < 64) gmax[tid] = init_bits;
    __syncthreads();

    const uint16_t *x = reinterpret_cast<const uint16_t *>(p.x) + xoff;
    const uint16_t *mean = nullptr;
    if (p.mean != nullptr)
        mean = reinterpret_cast<const uint16_t *>(p.mean) + (long)b * p.mean_sb + (long)h * p.mean_sh;

    float v[NCH][16];
    int rows[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = tid + 256 * i;
        const int r = c / CPR, col = (c % CPR) * 16;
        const int row = blk * BLK + r;
        rows[i] = row;
        float amax = 0.0f;
        if (row < L) {
            const v4u *src = reinterpret_cast<const v4u *>(x + (long)row * p.x_sl + col);
            v4u raw[2] = {src[0], src[1]};
            v4u mraw[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            if (mean != nullptr) {
                const v4u *ms = reinterpret_cast<const v4u *>(mean + col);
                mraw[0] = ms[0];
                mraw[1] = ms[1];
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const unsigned w = raw[j >> 3][(j & 7) >> 1];
                float f = ld16<DT>((uint16_t)((j & 1) ? (w >> 16) : (w & 0xffffu)));
                if (mean != nullptr) {
                    const unsigned mw = mraw[j >> 3][(j & 7) >> 1];
                    f = f - ld16<DT>((uint16_t)((j & 1) ? (mw >> 16) : (mw & 0xffffu)));
                    // torch's `k - km` rounds to the input dtype (quant_per_block.py:53-54);
                    // the CUDA quantiser keeps the fp32 difference (fused.cu:131-137)
                    if (p.style != QS_CUDA) f = ld16<DT>(st16<DT>(f));
                }
                f *= p.pre_scale;
                v[i][j] = f;
                amax = fmaxf(amax, fabsf(f));
            }
            atomicMax(&gmax[group_of_row(r, p.gran, p.warp)], __float_as_uint(amax));
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) v[i][j] = 0.0f;
        }
    }
    __syncthreads();

    if (tid < ngroups) sc_out[tid] = quant_scale(__uint_as_float(gmax[tid]), p.style);

#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = tid + 256 * i;
        const int r = c / CPR, col = (c % CPR) * 16;
        const int row = rows[i];
        if (row >= L) continue;
        const float am = __uint_as_float(gmax[group_of_row(r, p.gran, p.warp)]);
        int q[16];
        if (p.style == QS_CUDA) {
            const float inv = 127.0f / am;                           // fused.cu:164
#pragma unroll
            for (int j = 0; j < 16; j++) q[j] = quant_round_cuda(v[i][j], inv);
        } else {
            // x / scale must be the correctly rounded IEEE quotient (the reference divides, quant_per_block.py:41;
            // the +-0.5 / truncate that follows makes a 1-ulp error visible in the int8): sage_quant_math.h
            const float sc = quant_scale(am, p.style);
            const float y = quant_recip(sc);
            if (p.style == QS_TRITON_THREAD) {
#pragma unroll
                for (int j = 0; j < 16; j++) q[j] = quant_round_triton_nz(v[i][j], sc, y);
            } else {
#pragma unroll
                for (int j = 0; j < 16; j++) q[j] = quant_round_triton(v[i][j], sc, y);
            }
        }
        v4u pk;
#pragma unroll
        for (int w = 0; w < 4; w++) pk[w] = pack_int8x4(q[4 * w], q[4 * w + 1], q[4 * w + 2], q[4 * w + 3]);
        __builtin_nontemporal_store(pk, reinterpret_cast<v4u *>(p.out + ooff + (long)row * p.o_sl + col));   // written once, read by a later kernel
    }
}

template <int D, int BLK>
static hipError_t launch_dt(const QuantParams &p, dim3 grid, hipStream_t s)
{
    if (p.dtype == DT_F16) hipLaunchKernelGGL((quant_int8_kernel<D, BLK, DT_F16>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((quant_int8_kernel<D, BLK, DT_BF16>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_quant_int8(const QuantParams &p, hipStream_t stream)
{
    const int nblk = (p.L + p.blk - 1) / p.blk;          // varlen: p.L = max_seqlen
    if (nblk <= 0 || p.B <= 0 || p.H <= 0) return hipSuccess;
    dim3 grid(nblk, p.H, p.B);
    if (p.D == 128 && p.blk == 128) return launch_dt<128, 128>(p, grid, stream);
    if (p.D == 128 && p.blk == 64) return launch_dt<128, 64>(p, grid, stream);
    if (p.D == 64 && p.blk == 128) return launch_dt<64, 128>(p, grid, stream);
    if (p.D == 64 && p.blk == 64) return launch_dt<64, 64>(p, grid, stream);
    return hipErrorInvalidValue;
}

}  // namespace sage
