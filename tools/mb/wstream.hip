// Microbenchmark: how fast can every CU stream the SAME small weight buffer (L2-resident), as a function of the bytes it
// keeps in flight -- (a) LDS-DMA (global_load_lds, 1 KB per wave-instruction into a ring), (b) plain 16-byte loads to
// registers.  One 512-thread workgroup per CU, 8 waves, each wave moves 1 KB per unit of 8 KB.
//   hipcc --offload-arch=gfx950 -O3 tools/mb/wstream.hip -o /tmp/wstream && /tmp/wstream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int DEPTH, bool SKEW>
__global__ __launch_bounds__(512, 2) void k_dma(const unsigned char *w, int nunits, int reps, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int u = SKEW ? (blockIdx.x * 7) % nunits : 0;
    int slot = 0;
    const unsigned char *src = w + wid * 1024 + lane * 16;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)u * 8192),
                                         (__attribute__((address_space(3))) void *)(lds + slot * 8192 + wid * 1024), 16, 0, 0);
        u = (u + 1 == nunits) ? 0 : u + 1;
        slot = (slot + 1 == DEPTH + 1) ? 0 : slot + 1;
    }
    unsigned acc = 0;
    const int total = reps * nunits;
    for (int i = 0; i < total; i++) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)u * 8192),
                                         (__attribute__((address_space(3))) void *)(lds + slot * 8192 + wid * 1024), 16, 0, 0);
        u = (u + 1 == nunits) ? 0 : u + 1;
        slot = (slot + 1 == DEPTH + 1) ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc = *reinterpret_cast<unsigned *>(lds + tid * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int DEPTH, bool SKEW, bool BARRIER>
__global__ __launch_bounds__(512, 2) void k_reg(const unsigned char *w, int nunits, int reps, unsigned *sink)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int u = SKEW ? (blockIdx.x * 7) % nunits : 0;
    const unsigned char *src = w + wid * 1024 + lane * 16;
    uint4 buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        buf[d] = *reinterpret_cast<const uint4 *>(src + (size_t)u * 8192);
        u = (u + 1 == nunits) ? 0 : u + 1;
    }
    unsigned acc = 0;
    const int total = reps * nunits / DEPTH;
    for (int i = 0; i < total; i++) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            acc ^= buf[d].x ^ buf[d].w;                                  // consume the oldest
            if (BARRIER) __builtin_amdgcn_s_barrier();
            buf[d] = *reinterpret_cast<const uint4 *>(src + (size_t)u * 8192);
            u = (u + 1 == nunits) ? 0 : u + 1;
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) acc ^= buf[d].y;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F>
static double run(F launch, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e-3;
}

int main()
{
    const int nunits = 264;                     // 2.1 MB: the six convolutions of a C = 128, k = 11 residual block
    const int reps = 8;
    unsigned char *w;
    unsigned *sink;
    CK(hipMalloc(&w, (size_t)nunits * 8192 + 65536));
    CK(hipMemset(w, 1, (size_t)nunits * 8192 + 65536));
    CK(hipMalloc(&sink, 64));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const double bytes = (double)reps * nunits * 8192;
    printf("%d CUs, every CU streams the same %.1f MB x %d, 8 waves x 1 KB per 8 KB unit\n", ncu, nunits * 8192 / 1e6, reps);
#define DMA(D, S)                                                                                                    \
    {                                                                                                                \
        CK(hipFuncSetAttribute((const void *)k_dma<D, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
        double t = run([&] { hipLaunchKernelGGL((k_dma<D, S>), dim3(ncu), dim3(512), (D + 1) * 8192, 0, w, nunits, reps, sink); }, 5); \
        printf("LDS-DMA  %2d units (%3d KB) in flight, %s: %6.1f GB/s per CU, %5.2f TB/s chip\n", D, D * 8, S ? "skewed " : "lockstep", \
               bytes / t / 1e9, bytes * ncu / t / 1e12);                                                             \
    }
#define REG(D, S, B)                                                                                                 \
    {                                                                                                                \
        double t = run([&] { hipLaunchKernelGGL((k_reg<D, S, B>), dim3(ncu), dim3(512), 0, 0, w, nunits, reps, sink); }, 5); \
        printf("register %2d units (%3d KB) in flight, %s%s: %6.1f GB/s per CU, %5.2f TB/s chip\n", D, D * 8, S ? "skewed " : "lockstep", \
               B ? ", barrier per unit" : "", bytes / t / 1e9, bytes * ncu / t / 1e12);                              \
    }
    DMA(1, false) DMA(2, false) DMA(3, false) DMA(4, false) DMA(8, false) DMA(16, false)
    DMA(2, true) DMA(4, true) DMA(8, true)
    REG(2, false, false) REG(4, false, false) REG(8, false, false) REG(16, false, false)
    REG(4, true, false) REG(8, true, false)
    REG(4, false, true) REG(8, false, true)
    return 0;
}
