// ubench7: where does the hardware dispatcher put workgroup `blockIdx.x` when a grid exactly fills the chip's residency slots
// (256 CUs x 2 workgroups of 256 threads with 80 KB LDS each)?  Prints, per XCD, which (se, cu) each in-XCD index lands on and which
// indices share a CU: the attention launcher's short-grid work order (sage_attn.hip, "work item") is built on the answer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(256) probe(unsigned *out, int spin)
{
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_readcyclecounter();
    smem[threadIdx.x] = (unsigned char)spin;
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = hw; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = (unsigned)(t0 >> 4); out[blockIdx.x * 4 + 3] = smem[3]; }
}
int main(int argc, char **argv)
{
    const int lds = 80 * 1024;
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int nwg : {512, 1024}) {
        unsigned *d; hipMalloc(&d, nwg * 16);
        probe<<<nwg, 256, lds>>>(d, 200000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(nwg * 4); hipMemcpy(h.data(), d, nwg * 16, hipMemcpyDeviceToHost);
        printf("== grid %d\n", nwg);
        unsigned tmin = ~0u; for (int i = 0; i < nwg; i++) tmin = h[i * 4 + 2] < tmin ? h[i * 4 + 2] : tmin;
        for (int x = 0; x < 8; x++) {
            std::map<unsigned, std::vector<int>> cu;        // (se, sh, cu) -> in-XCD indices
            printf("xcd-by-blockIdx %d:", x);
            for (int idx = 0; idx * 8 + x < nwg; idx++) {
                const unsigned hw = h[(idx * 8 + x) * 4], xc = h[(idx * 8 + x) * 4 + 1] & 0xf;
                const unsigned cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                cu[(se << 8) | (sh << 4) | cuid].push_back(idx);
                if (x == 0 && idx < 72) printf(" %d:x%u.se%u.cu%u@%u", idx, xc, se, cuid, (h[(idx * 8 + x) * 4 + 2] - tmin) >> 6);
            }
            printf("\n  CUs used %zu; sharing:", cu.size());
            for (auto &kv : cu) { printf(" ["); for (int i : kv.second) printf("%d ", i); printf("]"); }
            printf("\n");
        }
        hipFree(d);
    }
    return 0;
}
