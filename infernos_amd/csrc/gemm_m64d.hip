// gemm_m64d.hip -- weight-streaming matrix product for decode steps of LLM-sized layers (17..64 rows, N x K in the tens of MB:
// Qwen2's gate|up 17920 x 1536 and the 151936-row vocabulary head; Cluster/InfernLLMWorker.py:103-119 -> generate()'s per-token
// forward), whole-line form of nn.hip's k_gemm_m64.
//
// k_gemm_m64 fetches every MFMA fragment straight into registers: a wave-instruction covers 16 rows x 64 bytes -- sixteen 128-byte
// lines for half of each, twice the address work per byte (what k_gemm_big paid until round 5) -- and it streams gate|up at 2.0 TB/s.
// Here a workgroup of four waves owns 64 columns (rows of w) x all (up to 64) rows of x and streams K in stages of 64: 8 units of
// 8 rows x 128 bytes of w and 8 of x per stage, each one global_load_lds wave-instruction fetching whole lines into a ring of four
// 16 KB slots (three stages in flight, two workgroups per CU: 96 KB on their way per CU); inside a unit the 16-byte chunk c of row
// r lies at chunk c ^ 2 (r / 2) (permutation on the source side), conflict-free for the ds_read_b128 fragment reads (the layout of
// gemm_big8.hip).  Wave w multiplies column tile w with the four row tiles: ONE ascending chain of 32-wide steps per output (k_gemm_m64:
// four chains over the quarters of K, summed through LDS) -- the last bits of an output differ from that kernel's, the epilogue
// arithmetic (folded normalisation, bias, SiLU gate, stores) is its own.  Arg-max keys (ifh_conv_desc.argmax_keys): per row the
// largest key of the workgroup's 64 columns through LDS, one atomic per row and workgroup.
#include <stdlib.h>

#include "igemm.h"

namespace ifh {

constexpr int M64D_SLOTS = 4, M64D_STAGE = 16 * 1024;

// ws != null: split-K form -- workgroup (x, z) multiplies stages [z per_z, (z + 1) per_z) and leaves its raw f32 partial tile in
// ws[z][M][N]; k_splitk_finish (nn.hip) adds the parts in z order and runs the epilogue (a deep narrow layer: the LLM's down projection,
// 1536 x 8960 -- 24 column blocks alone would leave 232 CUs idle)
__global__ __launch_bounds__(256, 2) void k_gemm_m64d(const IgemmParams p, float *__restrict__ ws, const int per_z)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 64;
    const int M = p.nbatch * p.T_out;
    const int nst_all = p.K / 64;
    const int s_beg = ws ? (int)blockIdx.y * per_z : 0;
    const int nst = ws ? min(per_z, nst_all - s_beg) : nst_all;

    // DMA units of this wave: w units 2 wid, 2 wid + 1 (rows n0 + 16 wid + 8 q + lane / 8), x units likewise (rows 16 wid + 8 q + lane / 8);
    // rows past N / M are clamped (their products are never stored)
    const int drow = lane >> 3, dch = (lane & 7) ^ ((drow >> 1) << 1);
    unsigned vw[2], vx[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int nr = min(n0 + 16 * wid + 8 * q + drow, p.N - 1);
        vw[q] = (unsigned)(((int64_t)nr * p.K) * 2 + dch * 16);
        const int mm = min(16 * wid + 8 * q + drow, M - 1);
        const int bb = mm / p.T_out, tt = mm - bb * p.T_out;
        vx[q] = (unsigned)(((int64_t)bb * p.x_bstride + (int64_t)tt * p.lda) * 2 + dch * 16);
    }
#define M64D_DMA(VOFF, BASE, DST)                                                                                         \
    do {                                                                                                                  \
        unsigned keep_;                                                                                                   \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(VOFF), "s"(BASE), "s"(DST) : "memory");                                        \
    } while (0)
    auto issue = [&](int stage, int slot) {
        const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.w) + (int64_t)(s_beg + stage) * 128;
        const unsigned char *xb = reinterpret_cast<const unsigned char *>(p.x) + (int64_t)(s_beg + stage) * 128;
        const int dst = slot * M64D_STAGE + (2 * wid) * 1024;
        M64D_DMA(vw[0], wb, dst);
        M64D_DMA(vw[1], wb, dst + 1024);
        M64D_DMA(vx[0], xb, dst + 8 * 1024);
        M64D_DMA(vx[1], xb, dst + 9 * 1024);
    };

    // fragment reads: row fr of a 16-row block = unit fr / 8, row fr % 8; k step h: chunks 4 h + fg at chunk ^ 2 (row / 2)
    const int rr8 = fr & 7, fsw = (rr8 >> 1) << 1;
    const int fo0 = (fr >> 3) * 1024 + rr8 * 128 + ((fg ^ fsw) << 4), fo1 = (fr >> 3) * 1024 + rr8 * 128 + (((4 + fg) ^ fsw) << 4);
    const unsigned char *a0 = lds + (2 * wid) * 1024 + fo0, *a1 = lds + (2 * wid) * 1024 + fo1;
    const unsigned char *b0 = lds + 8 * 1024 + fo0, *b1 = lds + 8 * 1024 + fo1;

    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // three stages ahead; a stage's four pieces per wave are waited for by count (younger: the stages behind it), one barrier per stage
    // publishes them and frees the slot of the stage before
#pragma unroll
    for (int s = 0; s < M64D_SLOTS - 1; s++)
        if (s < nst) issue(s, s);
    for (int s = 0; s < nst; s++) {
        const int ahead = min(M64D_SLOTS - 2, nst - 1 - s);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + M64D_SLOTS - 1 < nst) issue(s + M64D_SLOTS - 1, (s + M64D_SLOTS - 1) % M64D_SLOTS);
        const int so = (s % M64D_SLOTS) * M64D_STAGE;
        const bf16x8_t wa = *reinterpret_cast<const bf16x8_t *>(a0 + so), wb_ = *reinterpret_cast<const bf16x8_t *>(a1 + so);
        bf16x8_t xa[4], xb_[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            xa[r] = *reinterpret_cast<const bf16x8_t *>(b0 + so + r * 2048);
            xb_[r] = *reinterpret_cast<const bf16x8_t *>(b1 + so + r * 2048);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xa[r], acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb_, xb_[r], acc[r], 0, 0, 0);
    }
#undef M64D_DMA

    // ---- epilogue (k_gemm_m64's): lane (fr, fg) holds columns n .. n + 3 of row m = 16 r + fr
    const int n = n0 + 16 * wid + 4 * fg;
    if (ws) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = 16 * r + fr;
            if (m < M && n < p.N)
                *reinterpret_cast<float4 *>(ws + ((int64_t)blockIdx.y * M + m) * p.N + n) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        }
        return;
    }
    unsigned long long best[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int m = 16 * r + fr;
        f32x4 s = acc[r];
        if (m < M && n < p.N) {
            const int dynv = dyn_value(p, m);
            if (p.aln_stats) {
                // normalisation of the A rows folded in (ifh_conv_desc.aln_*): the producer left (sum, sum of squares)
                const longlong2 st = reinterpret_cast<const longlong2 *>(p.aln_stats)[m];
                const float fx = (1.0f / 65536.0f) / (float)p.ln_dim;
                const float mean = p.ln_rms ? 0.0f : (float)st.x * fx;
                const float rstd = rsqrtf(fmaxf((float)st.y * fx - mean * mean, 0.0f) + p.ln_eps);
                float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!p.ln_rms) c1 = *reinterpret_cast<const float4 *>(p.aln_c1 + n);
                s[0] = rstd * (s[0] - mean * c1.x);
                s[1] = rstd * (s[1] - mean * c1.y);
                s[2] = rstd * (s[2] - mean * c1.z);
                s[3] = rstd * (s[3] - mean * c1.w);
            }
            if (p.amax_keys) {
                // (the launcher admits no bias / activation / residual / scale here: s is what igemm_store4 stores)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (n + e < p.N) {
                        const uint32_t u = __float_as_uint(s[e]);
                        const uint32_t ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                        const unsigned long long key = ((unsigned long long)ord << 32) | (uint32_t)(0xffffffffu - (uint32_t)(n + e));
                        best[r] = key > best[r] ? key : best[r];
                    }
            }
            if (p.act == ACT_SILU_GLU) {
                // interleaved (gate, up) weight rows: this lane holds two pairs -> two outputs of the half-width result
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4 *>(p.bias + n);
                    s[0] += bv.x; s[1] += bv.y; s[2] += bv.z; s[3] += bv.w;
                }
                const float o0 = s[0] / (1.0f + __expf(-s[0])) * s[1], o1 = s[2] / (1.0f + __expf(-s[2])) * s[3];
                *reinterpret_cast<uint32_t *>(reinterpret_cast<uint16_t *>(p.out) + (int64_t)m * p.ldc + (n >> 1)) = f32x2_to_bf16x2(o0, o1);
            } else if (p.fast_epi)
                igemm_store4<true>(p, m, n, s, dynv);
            else
                igemm_store4<false>(p, m, n, s, dynv);
        }
    }
    if (p.amax_keys) {
        // a row's largest key over the workgroup's 64 columns: over the four lanes of a wave that share the row, then over the
        // waves through LDS (the ring is idle); one atomic per row and workgroup
        unsigned long long *red = reinterpret_cast<unsigned long long *>(lds);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            unsigned long long b = best[r], o = __shfl_xor(b, 16, 64);
            b = o > b ? o : b;
            o = __shfl_xor(b, 32, 64);
            b = o > b ? o : b;
            if (fg == 0) red[wid * 64 + 16 * r + fr] = b;
        }
        __syncthreads();
        if (tid < M) {
            unsigned long long b = red[tid];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                const unsigned long long o = red[w * 64 + tid];
                b = o > b ? o : b;
            }
            if (b) atomicMax(p.amax_keys + tid, b);
        }
    }
}

static bool m64d_ready(const IgemmParams &p, int64_t M)
{
    static const int on = getenv("IFH_GEMM_M64D") ? atoi(getenv("IFH_GEMM_M64D")) : 1;      // tuning switch: 0 = the register-fragment kernels
    if (!on || p.K % 64 || p.K < 64 || (((uintptr_t)p.x) & 15) || (((uintptr_t)p.w) & 15) || p.lda % 8 || p.x_bstride % 8 || !p.vec_ok) return false;
    if ((int64_t)p.N * p.K * 2 >= (1ll << 32) || ((int64_t)(p.nbatch - 1) * p.x_bstride + (int64_t)p.T_out * p.lda) * 2 >= (1ll << 32)) return false;
    if (M < 1 || M > 64) return false;
    constexpr int bytes = M64D_SLOTS * M64D_STAGE;
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        if (hipFuncSetAttribute((const void *)k_gemm_m64d, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
        attr_once.done(attr_dev);
    }
    return true;
}

// true if it took the launch: nn.hip has checked what k_gemm_m64 takes (17..64 rows, one tap, wide layer, epilogue); here: K in 64s,
// 16-byte addressable operands, 32-bit DMA offsets
bool try_launch_gemm_m64d(const IgemmParams &p, hipStream_t st)
{
    const int64_t M = (int64_t)p.nbatch * p.T_out;
    if (!m64d_ready(p, M)) return false;
    hipLaunchKernelGGL(k_gemm_m64d, dim3((unsigned)((p.N + 63) / 64)), dim3(256), M64D_SLOTS * M64D_STAGE, st, p, (float *)nullptr, 0);
    return true;
}

// the split-K form: the number of parts (0 = not taken) -- as many as give every CU two workgroups, at least four stages each, as the
// caller's workspace holds (ws_floats >= parts * rows * n); the caller then runs k_splitk_finish over `parts`
int try_launch_gemm_m64d_splitk(const IgemmParams &p, float *ws, int64_t ws_floats, hipStream_t st)
{
    const int64_t M = (int64_t)p.nbatch * p.T_out;
    if (!ws || (((uintptr_t)ws) & 15) || p.N % 4 || !m64d_ready(p, M)) return 0;
    const int nb = (p.N + 63) / 64, nst = p.K / 64;
    int parts = (2 * device_cu_count_physical() + nb - 1) / nb;
    if (parts > nst / 4) parts = nst / 4;
    if ((int64_t)parts * M * p.N > ws_floats) parts = (int)(ws_floats / (M * p.N));
    if (parts < 2) return 0;
    const int per_z = (nst + parts - 1) / parts;
    parts = (nst + per_z - 1) / per_z;                   // no empty part
    hipLaunchKernelGGL(k_gemm_m64d, dim3((unsigned)nb, (unsigned)parts), dim3(256), M64D_SLOTS * M64D_STAGE, st, p, ws, per_z);
    return parts;
}

}  // namespace ifh
