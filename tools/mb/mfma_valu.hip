// What a vector instruction costs beside v_mfma_f32_16x16x32_bf16: every wave runs  { MFMA ; NV x <vector op> }  x 16 per iteration
// (pure registers, no memory), one or two waves per SIMD, and reports shader clocks per MFMA of wave 0.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu tools/mb/mfma_valu.hip && ./mfma_valu
// Vector ops (inline asm, independent of the MFMAs and of each other, 8 registers in rotation):
//   kind 0  v_add_f32        kind 1  v_pk_mul_f32        kind 2  v_cvt_pk_bf16_f32        kind 3  v_and_b32
// Reading: "additive" = 16 + 4 NV clocks per MFMA with one wave per SIMD (twice that per wave with two); "hidden" = max(16, 8 + 4 NV).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int KIND, int NACC>
__global__ __launch_bounds__(512) void k_mix(float *out, unsigned long long *clk, int iters, uint32_t seed)
{
    uint4 ua, ub;
    const uint32_t t = threadIdx.x * 2654435761u + blockIdx.x * 40503u + seed;
    ua = make_uint4(t * 3u | 0x3f803f80u, (t >> 3) * 7u, t * 11u, (t >> 5) * 13u);
    ub = make_uint4(t * 17u, (t >> 2) * 19u, t * 23u, (t >> 7) * 29u);
    ua.x &= 0x3fff3fffu; ua.y &= 0x3fff3fffu; ua.z &= 0x3fff3fffu; ua.w &= 0x3fff3fffu;
    ub.x &= 0x3fff3fffu; ub.y &= 0x3fff3fffu; ub.z &= 0x3fff3fffu; ub.w &= 0x3fff3fffu;
    const bf16x8_t a = __builtin_bit_cast(bf16x8_t, ua), b = __builtin_bit_cast(bf16x8_t, ub);
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x2 f[8];
    uint32_t u[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        f[j] = (f32x2){(float)(t & 255) * 1e-3f + j, 1.0f + j * 1e-3f};
        u[j] = t + j;
    }
    const float c = 1.0000001f;
    const f32x2 c2 = {1.0000001f, 0.9999999f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                const int j = (i * NV + v) & 7;
                if (KIND == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[j].x) : "v"(c));
                if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(f[j]) : "v"(c2));
                if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[j]) : "v"(f[j].x), "v"(f[j].y));
                if (KIND == 3) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[j]) : "v"(0xfffffff7u + v));
                // kinds 4 / 5 / 6: NV plain ops, then TWO packed ones (does a packed op pay its 12 extra clocks when plain ops sit between
                // it and the MFMA?); kind 7: two packed ones FIRST, then NV plain ops
                if (KIND >= 4 && KIND <= 6) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[j]) : "v"(0xfffffff7u + v));
                if (KIND == 7 && v == 0) {
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(f[0]) : "v"(c2));
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(f[1]) : "v"(c2));
                }
                if (KIND == 7) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[j]) : "v"(0xfffffff7u + v));
                if (KIND >= 4 && KIND <= 6 && v == NV - 1) {
                    if (KIND == 4) {
                        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(f[0]) : "v"(c2));
                        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(f[1]) : "v"(c2));
                    }
                    if (KIND == 5) {
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(c2));
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f[1]) : "v"(c2));
                    }
                    if (KIND == 6) {       // the same arithmetic as four plain multiplies
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[0].x) : "v"(c));
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[0].y) : "v"(c));
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[1].x) : "v"(c));
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[1].y) : "v"(c));
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int j = 0; j < 8; j++) s += f[j].x + f[j].y + (float)u[j];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NV, int KIND, int NACC>
static void run(int waves, float *out, unsigned long long *clk, const char *kname)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<NV, KIND, NACC>), dim3(256), dim3(waves * 64), 0, 0, out, clk, 200, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<NV, KIND, NACC>), dim3(256), dim3(waves * 64), 0, 0, out, clk, iters, 2u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h = 0;
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / ((double)iters * 16);
    printf("%-18s acc %2d  waves/SIMD %d  NV %d: %6.1f clocks per MFMA of a wave  (%.2f GHz, %.0f TFLOP/s)\n", kname, NACC, waves / 4, NV, per,
           (double)h / (ms * 1e6), 256.0 * waves * iters * 16 * 16384.0 / (ms * 1e-3) / 1e12);
}

#define RUN_ALL(KIND, NAME, NACC)                          \
    for (int w = 4; w <= 8; w += 4) {                      \
        run<0, KIND, NACC>(w, out, clk, NAME);             \
        run<1, KIND, NACC>(w, out, clk, NAME);             \
        run<2, KIND, NACC>(w, out, clk, NAME);             \
        run<3, KIND, NACC>(w, out, clk, NAME);             \
        run<4, KIND, NACC>(w, out, clk, NAME);             \
        run<6, KIND, NACC>(w, out, clk, NAME);             \
        run<8, KIND, NACC>(w, out, clk, NAME);             \
    }

int main(int argc, char **argv)
{
    (void)argv;
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, 4);
    hipMalloc(&clk, 8);
    if (argc > 1) {          // the second table: packed ops behind NV plain ones
        RUN_ALL(4, "NV and + 2 pk_mul", 16)
        RUN_ALL(5, "NV and + 2 pk_add", 16)
        RUN_ALL(6, "NV and + 4 mul", 16)
        RUN_ALL(7, "2 pk_mul + NV and", 16)
        return 0;
    }
    RUN_ALL(0, "v_add_f32", 16)
    RUN_ALL(0, "v_add_f32", 2)
    RUN_ALL(1, "v_pk_mul_f32", 16)
    RUN_ALL(2, "v_cvt_pk_bf16_f32", 16)
    RUN_ALL(3, "v_and_b32", 16)
    return 0;
}
