// Pure-register MFMA rate of the box: every wave issues independent v_mfma_f32_16x16x32_bf16 on 16 accumulators, no memory.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/mb/mfma_peak.hip && ./mfma_peak [waves_per_block] [ms]
// Operands are random-looking bit patterns (zero operands toggle fewer wires and read high).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_mfma(float *out, int iters, uint32_t seed)
{
    uint4 ua, ub;
    const uint32_t t = threadIdx.x * 2654435761u + blockIdx.x * 40503u + seed;
    ua = make_uint4(t * 3u | 0x3f803f80u, (t >> 3) * 7u, t * 11u, (t >> 5) * 13u);
    ub = make_uint4(t * 17u, (t >> 2) * 19u, t * 23u, (t >> 7) * 29u);
    // keep exponents moderate: clear the top exponent bits of every bf16
    ua.x &= 0x3fff3fffu; ua.y &= 0x3fff3fffu; ua.z &= 0x3fff3fffu; ua.w &= 0x3fff3fffu;
    ub.x &= 0x3fff3fffu; ub.y &= 0x3fff3fffu; ub.z &= 0x3fff3fffu; ub.w &= 0x3fff3fffu;
    const bf16x8_t a = __builtin_bit_cast(bf16x8_t, ua), b = __builtin_bit_cast(bf16x8_t, ub);
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char **argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 8;
    const double target_ms = argc > 2 ? atof(argv[2]) : 50.0;
    float *out;
    hipMalloc(&out, 4);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int iters = 2000;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, (uint32_t)rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * waves * iters * 16.0 * (2.0 * 16 * 16 * 32);
        printf("waves/block %d blocks %d iters %d: %.2f ms -> %.0f TFLOP/s\n", waves, blocks, iters, ms, flop / ms / 1e9);
        if (rep == 0) iters = (int)(iters * target_ms / (ms > 0.01 ? ms : 0.01));
    }
    return 0;
}
