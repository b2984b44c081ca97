// ubench9: ds_read_b64_tr_b16 as the way to feed the PV MFMA's A operand (V^T) from ROW-MAJOR V rows in LDS -- round 6, the question behind
// "no V image for fp16 inputs" (DESIGN.md 8-1).
//  (1) semantics: 64 lanes read the 8-byte chunk at lane * 8; which (source lane, element) does each of a lane's four results come from?
//      Hypothesis H: inside every group of 16 lanes, out[j] of lane i = element (i & 3) of the chunk supplied by lane 4 j + (i >> 2).
//  (2) a 64-token x D fp16 tile stored as rows (row = D * 2 bytes, 64-byte segments XOR-swizzled by the row so that the four rows a 32-lane
//      half touches fall into different banks): is the operand (lane = channel, 8 tokens 16c + 8 (j >> 2) + 4 g + (j & 3)) what two transposing
//      reads return, and what does a tile's worth of them cost against the 16 ds_read_b128 of the pre-transposed image?
// build: hipcc -O3 --offload-arch=gfx950 ubench9_tr_b16.hip -o ubench9
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2u tr_read(unsigned addr)
{
    v2u r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    return r;
}

__global__ void probe_semantics(uint16_t *out)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t *)lds;
    const v2u r = tr_read(base + lane * 8);
    out[4 * lane + 0] = (uint16_t)(r[0] & 0xffffu);
    out[4 * lane + 1] = (uint16_t)(r[0] >> 16);
    out[4 * lane + 2] = (uint16_t)(r[1] & 0xffffu);
    out[4 * lane + 3] = (uint16_t)(r[1] >> 16);
}

// byte offset of (token row t, channel d) in the row-major tile: 64-byte segments XOR-swizzled by the row
template <int D> __device__ __host__ inline unsigned row_off(int t, int d, int swz)
{
    constexpr int ROWB = D * 2;
    unsigned seg = (unsigned)(d * 2) >> 6, in = (unsigned)(d * 2) & 63u;
    if (swz) seg ^= (D == 128) ? (unsigned)(t & 3) : (unsigned)((t >> 1) & 1);
    return (unsigned)t * ROWB + seg * 64 + in;
}

template <int D>
__global__ void operand_check(uint16_t *out, int swz)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    // element (t, d) = t * 256 + d (unique for t < 64, d < 256)
    for (int e = threadIdx.x; e < 64 * D; e += blockDim.x) {
        const int t = e / D, d = e % D;
        *reinterpret_cast<uint16_t *>(smem + row_off<D>(t, d, swz)) = (uint16_t)(t * 256 + d);
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int m = lane & 31, g = lane >> 5;
    // the lane SUPPLIES (hypothesis H): inside its 16-lane group it is source s = lane & 15: row t0 + (s >> 2), channels dbase + 4 (s & 3) .. + 3
    const int s = lane & 15, dbase16 = 16 * ((lane >> 4) & 1);
    for (int dt = 0; dt < D / 32; dt++)
        for (int c = 0; c < 4; c++)
            for (int half = 0; half < 2; half++) {
                const int t0 = 16 * c + 8 * half + 4 * g;
                const unsigned a = base + row_off<D>(t0 + (s >> 2), 32 * dt + dbase16 + 4 * (s & 3), swz);
                const v2u r = tr_read(a);
                uint16_t *o = out + ((((dt * 4 + c) * 2 + half) * 64 + lane) * 4);
                o[0] = (uint16_t)(r[0] & 0xffffu); o[1] = (uint16_t)(r[0] >> 16); o[2] = (uint16_t)(r[1] & 0xffffu); o[3] = (uint16_t)(r[1] >> 16);
                (void)m;
            }
}

// timing: MODE 0 = 32 (D = 128) transposing reads per tile from rows, MODE 1 = 16 ds_read_b128 from the image layout (swizzled as the product's)
template <int D, int MODE>
__global__ void __launch_bounds__(256) read_cost(unsigned *sink, long long *cycles, int iters, int swz)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int e = threadIdx.x; e < 64 * D / 2; e += blockDim.x) reinterpret_cast<unsigned *>(smem)[e] = e * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 5, n = lane & 31, s = lane & 15, dbase16 = 16 * ((lane >> 4) & 1);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned acc = 0;
    unsigned addr[D / 32][4][2];
    if (MODE == 0) {
        for (int dt = 0; dt < D / 32; dt++)
            for (int c = 0; c < 4; c++)
                for (int half = 0; half < 2; half++)
                    addr[dt][c][half] = base + row_off<D>(16 * c + 8 * half + 4 * g + (s >> 2), 32 * dt + dbase16 + 4 * (s & 3), swz);
    }
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int dt = 0; dt < D / 32; dt++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    v2u r0, r1;
                    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(r0), "=&v"(r1) : "v"(addr[dt][c][0]), "v"(addr[dt][c][1]) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    acc ^= r0[0] ^ r0[1] ^ r1[0] ^ r1[1];
                }
        } else {
#pragma unroll
            for (int dt = 0; dt < D / 32; dt++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int drow = dt * 32 + n;
                    const int chunk = (4 * g + c) ^ ((drow >> 1) & 7);
                    v4u r;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(base + drow * 128 + chunk * 16) : "memory");
                    acc ^= r[0] ^ r[1] ^ r[2] ^ r[3];
                }
        }
    }
    const long long t1 = clock64();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int D> static int check_operand(int swz)
{
    const int n = (D / 32) * 4 * 2 * 64 * 4;
    uint16_t *d_out;
    hipMalloc(&d_out, n * 2);
    hipLaunchKernelGGL(operand_check<D>, dim3(1), dim3(256), 64 * D * 2, 0, d_out, swz);
    std::vector<uint16_t> h(n);
    hipMemcpy(h.data(), d_out, n * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int dt = 0; dt < D / 32; dt++)
        for (int c = 0; c < 4; c++)
            for (int half = 0; half < 2; half++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 4; j++) {
                        const int m = lane & 31, g = lane >> 5;
                        const int t = 16 * c + 8 * half + 4 * g + j, d = 32 * dt + m;
                        const uint16_t want = (uint16_t)(t * 256 + d), got = h[((((dt * 4 + c) * 2 + half) * 64 + lane) * 4) + j];
                        if (want != got && bad++ < 4) printf("  D=%d swz=%d dt=%d c=%d half=%d lane=%d j=%d: want (t %d, d %d) got (t %d, d %d)\n", D, swz, dt, c, half, lane, j, t, d, got / 256, got % 256);
                    }
    hipFree(d_out);
    printf("operand from row-major V tile, D=%d swizzle=%d: %s (%d wrong)\n", D, swz, bad ? "WRONG" : "ok", bad);
    return bad;
}

template <int D, int MODE> static void time_reads(int swz, int wgs_per_cu)
{
    const int blocks = 256 * wgs_per_cu, iters = 2000;
    unsigned *sink; long long *cyc;
    hipMalloc(&sink, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL((read_cost<D, MODE>), dim3(blocks), dim3(256), 64 * D * 2, 0, sink, cyc, 10, swz);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((read_cost<D, MODE>), dim3(blocks), dim3(256), 64 * D * 2, 0, sink, cyc, iters, swz);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : h) avg += (double)x; avg /= blocks;
    printf("  D=%d %-28s swz=%d, %d wg/CU: %7.1f clock64 ticks per tile per wave (%6.1f ns per tile by wall time)\n", D,
           MODE == 0 ? "32/16 x ds_read_b64_tr_b16 rows" : "16/8 x ds_read_b128 image", swz, wgs_per_cu, avg / iters, ms * 1e6 / iters);
    hipFree(sink); hipFree(cyc);
}

int main()
{
    uint16_t *d_out;
    hipMalloc(&d_out, 256 * 2);
    hipLaunchKernelGGL(probe_semantics, dim3(1), dim3(64), 0, 0, d_out);
    uint16_t h[256];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    printf("ds_read_b64_tr_b16 with lane address = lane * 8: out[j] of lane i as (source lane, element)\n");
    for (int lane = 0; lane < 64; lane++) {
        if (lane < 20 || lane == 63) printf("  lane %2d:", lane);
        for (int j = 0; j < 4; j++) {
            const int v = h[4 * lane + j], src = v / 4, el = v % 4;
            const int i = lane & 15, grp = lane >> 4;
            const int want_src = 16 * grp + 4 * j + (i >> 2), want_el = i & 3;
            if (src != want_src || el != want_el) bad++;
            if (lane < 20 || lane == 63) printf(" (%2d,%d)", src, el);
        }
        if (lane < 20 || lane == 63) printf("\n");
    }
    printf("hypothesis H (out[j] of lane i = element (i & 3) of the chunk of lane 16 grp + 4 j + (i >> 2)): %s\n", bad ? "WRONG" : "holds");
    for (int swz = 0; swz < 2; swz++) { check_operand<128>(swz); check_operand<64>(swz); }
    printf("cost of one 64-token tile's PV operand reads per wave (4 waves per workgroup, every read waited for):\n");
    for (int wg = 1; wg <= 2; wg++) {
        time_reads<128, 1>(1, wg);
        time_reads<128, 0>(0, wg);
        time_reads<128, 0>(1, wg);
        time_reads<64, 1>(1, wg);
        time_reads<64, 0>(0, wg);
        time_reads<64, 0>(1, wg);
    }
    return 0;
}
