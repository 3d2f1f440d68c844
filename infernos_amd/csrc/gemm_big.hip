// gemm_big.hip -- out[M][N] = act(x[M][K] @ w[N][K]^T + bias) (+ resid), bf16 in / f32 accumulate / bf16 out, for the
// matrix products with thousands of rows: the Whisper / SpeechT5 encoder layers (Cluster/InfernSTTWorker.py:65 ->
// WhisperEncoderLayer, 128 windows x 1500 positions = 192 000 rows; HelloSippyRTPipe.py:47-110's text encoder), LLM prefill.
//
// k_igemm<128,128> (nn.hip) runs these at 530-810 TFLOP/s: a 128 x 128 tile moves 64 FLOP per L2 byte, its k-loop is a
// load -> barrier round trip per 32..64 of K that only other resident workgroups cover, and its epilogue (GELU on 64 outputs per
// thread at fc1) runs with the matrix pipe idle.  Here
//   * a workgroup of FOUR waves owns a 256 (columns of w) x 128 (rows of x) tile, 128 x 64 per wave = 32 accumulator tiles:
//     85 FLOP per L2 byte, 12 fragment reads per 32 MFMAs (94 B/clk of LDS reads at full rate);
//   * operands go global -> LDS by DMA (global_load_lds, 16 B per lane, no VGPR round trip) into a ring of three 32-deep K
//     stages; a stage is 24 blocks of 16 rows x 64 B = one MFMA fragment each, swizzled on the SOURCE side (the DMA writes LDS
//     linearly) so that ds_read_b128 fragment reads are conflict-free; two stages stay in flight across the ONE barrier per
//     stage (counted vmcnt, raw s_barrier: the DMA is inline asm, hipcc does not see it and so never drains it);
//   * 72 KB of LDS and <= 256 VGPRs: TWO workgroups per CU that drift apart by themselves -- one's epilogue (bias, GELU,
//     residual, stores) and ring fill run under the other's MFMAs, so nothing has to be persistent or software-pipelined
//     across tiles;
//   * workgroups that run at the same time on one XCD take the column tiles of the same rows of x (one L2 fetch of x).
// Same k order per output element (ascending 32-wide steps of v_mfma_f32_16x16x32_bf16 into one chain) and the same epilogue
// arithmetic as k_igemm: the same bits.
#include <stdlib.h>

#include "igemm.h"

namespace ifh {

bool try_launch_gemm_big8(const IgemmParams &p, int64_t M, hipStream_t st, float *ws, int64_t ws_floats);

struct GemmBigParams {
    const uint16_t *x;
    int lda;
    const uint16_t *w;           // [N][K]
    const float *bias;           // [N] or null
    const uint16_t *resid;       // [M][ldr] or null
    int ldr;
    uint16_t *out;
    int ldc;
    int M, N, K;
    int mtiles, ntiles;
    int abl;                     // diagnostic (IFH_GEMM_BIG_ABL, wrong results): 1 = no DMA after the first two stages, 2 = no MFMAs, 4 = no epilogue
};

constexpr int GB_BN = 256, GB_BM = 128, GB_BK = 32, GB_STAGES = 3;
constexpr int GB_ABLK = GB_BN / 16, GB_BBLK = GB_BM / 16, GB_BLKS = GB_ABLK + GB_BBLK;      // 16 + 8 blocks of 1 KB per stage
constexpr int GB_STAGE_BYTES = GB_BLKS * 1024;

template <int ACT>
__global__ __launch_bounds__(256, 2) void k_gemm_big(const GemmBigParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid & 1, wm = wid >> 1;             // wave: columns [128 wn, +128) x rows [64 wm, +64) of the tile
    const int fr = lane & 15, fg = lane >> 4;

    // tile of this workgroup: blocks b and b + 8 run on one XCD; give an XCD a contiguous run of tiles, column tile fastest
    const int total = p.mtiles * p.ntiles;
    int t = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = t & 7, i = t >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int mt = t / p.ntiles, nt = t - mt * p.ntiles;
    const int n0 = nt * GB_BN, m0 = mt * GB_BM;

    // ---- DMA: block j of a stage (j < 16: rows n0 + 16 j .. of w; else rows m0 + 16 (j - 16) .. of x) is one wave-instruction:
    // lane l fetches 16 bytes of row l / 4, 16-byte chunk (l % 4) ^ swz(l / 4) -- the swizzle on the source side
    const int drow = lane >> 2, dch = (lane & 3) ^ (((drow >> 2) & 1) << 1);
    unsigned voff[GB_BLKS / 4];                        // this wave's six blocks: j = wid + 4 q
#pragma unroll
    for (int q = 0; q < GB_BLKS / 4; q++) {
        const int j = wid + 4 * q;
        voff[q] = j < GB_ABLK ? (unsigned)(((n0 + j * 16 + drow) * p.K + dch * 8) * 2)
                              : (unsigned)(((m0 + (j - GB_ABLK) * 16 + drow) * p.lda + dch * 8) * 2);
    }
    auto issue_stage = [&](int stage_k, int buf) {     // K range [32 stage_k, +32) into ring buffer buf
        const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.w) + (int64_t)stage_k * (GB_BK * 2);
        const unsigned char *xb = reinterpret_cast<const unsigned char *>(p.x) + (int64_t)stage_k * (GB_BK * 2);
#pragma unroll
        for (int q = 0; q < GB_BLKS / 4; q++) {
            const int j = wid + 4 * q;
            const unsigned char *base = j < GB_ABLK ? wb : xb;        // wave-uniform
            const int dst = buf * GB_STAGE_BYTES + j * 1024;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff[q]), "s"(base), "s"(dst) : "memory");
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nstages = p.K / GB_BK;
    issue_stage(0, 0);
    if (nstages > 1) issue_stage(1, 1);
    // fragment read offset of this lane inside a block: row fr, chunk fg ^ swz(fr)
    const int foff = fr * 64 + ((fg ^ (((fr >> 2) & 1) << 1)) << 4);
    const int aoff = (wn * 8) * 1024 + foff, boff = (GB_ABLK + wm * 4) * 1024 + foff;
    // Stage s: its DMA pieces are waited for, one barrier publishes them, its twelve fragment reads go out, the DMA of stage s + 2 is
    // issued in their shadow (into the buffer stage s - 1 has left), then the 32 MFMAs follow.  (A second fragment register set, so
    // that a stage's reads run under the previous stage's MFMAs, spilled: 128 accumulator + 96 fragment registers.)
    for (int s = 0; s < nstages; s++) {
        if (p.abl & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (s + 1 < nstages) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // stage s + 1's six pieces may still be on their way
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave's pieces of stage s are in LDS; nobody reads stage s - 1's buffer again
        const unsigned char *st = lds + (s % GB_STAGES) * GB_STAGE_BYTES;
        bf16x8_t fa[8], fb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j] = *reinterpret_cast<const bf16x8_t *>(st + boff + j * 1024);
#pragma unroll
        for (int i = 0; i < 8; i++) fa[i] = *reinterpret_cast<const bf16x8_t *>(st + aoff + i * 1024);
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);            // the DS reads first
        if (s + 2 < nstages && !(p.abl & 1)) issue_stage(s + 2, (s + 2) % GB_STAGES);
        if (!(p.abl & 2)) {
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        } else {
            asm volatile("" ::"v"(fa[0]), "v"(fa[7]), "v"(fb[0]), "v"(fb[3]));
        }
    }
    if (p.abl & 4) return;

    // ---- epilogue.  D[n][m]: a lane holds 4 consecutive columns n of one row m -- stored as they lie, a wave-instruction writes
    // 16 rows x 32 bytes (quarter lines: the stores alone were 44 % of the qkv launch).  So the wave's 64 x 128 sub-tile goes through
    // LDS (the ring is idle now; 272-byte rows: 16-byte aligned, 2-way on the 8-byte writes) and leaves as whole 256-byte rows,
    // 16 bytes per lane; the residual comes in the same way and is added in f32 before the one rounding, as k_igemm does.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // nobody reads the ring any more
    constexpr int EROW = 272;
    unsigned char *wbuf = lds + wid * (64 * EROW);
    const int mrow0 = m0 + wm * 64, ncol0 = n0 + wn * 128;
    const int trow = lane >> 4, tch = lane & 15;
    if (p.resid) {
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const int row = it * 4 + trow;
            *reinterpret_cast<uint4 *>(wbuf + row * EROW + tch * 16) =
                *reinterpret_cast<const uint4 *>(p.resid + (int64_t)(mrow0 + row) * p.ldr + ncol0 + tch * 8);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int n = ncol0 + i * 16 + 4 * fg;
            unsigned char *slot = wbuf + (j * 16 + fr) * EROW + (i * 16 + 4 * fg) * 2;
            float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
            if (p.bias) {
                const float4 bv = *reinterpret_cast<const float4 *>(p.bias + n);
                v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
            }
            if (ACT == ACT_SILU_GLU) {
                // interleaved (gate, up) weight rows: the lane's four columns are two pairs -> two outputs of the half-width result
                // (k_gemm_m64's arithmetic, nn.hip); they go to the first half of the transposition row
                const float o0 = v0 / (1.0f + __expf(-v0)) * v1, o1 = v2 / (1.0f + __expf(-v2)) * v3;
                *reinterpret_cast<uint32_t *>(wbuf + (j * 16 + fr) * EROW + (i * 16 + 4 * fg)) = f32x2_to_bf16x2(o0, o1);
                continue;
            }
            if (ACT != ACT_NONE) {
                v0 = apply_act_c<ACT>(v0, ACT, 0.0f); v1 = apply_act_c<ACT>(v1, ACT, 0.0f);
                v2 = apply_act_c<ACT>(v2, ACT, 0.0f); v3 = apply_act_c<ACT>(v3, ACT, 0.0f);
            }
            if (p.resid) {
                const uint2 rv = *reinterpret_cast<const uint2 *>(slot);
                v0 += __uint_as_float(rv.x << 16);
                v1 += __uint_as_float(rv.x & 0xffff0000u);
                v2 += __uint_as_float(rv.y << 16);
                v3 += __uint_as_float(rv.y & 0xffff0000u);
            }
            *reinterpret_cast<uint2 *>(slot) = make_uint2(f32x2_to_bf16x2(v0, v1), f32x2_to_bf16x2(v2, v3));
        }
    }
    if (ACT == ACT_SILU_GLU) {                         // 64 outputs per row of the sub-tile: 128 bytes, eight 16-byte pieces
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const int row = it * 4 + trow;
            if (tch < 8)
                *reinterpret_cast<uint4 *>(p.out + (int64_t)(mrow0 + row) * p.ldc + (ncol0 >> 1) + tch * 8) =
                    *reinterpret_cast<const uint4 *>(wbuf + row * EROW + tch * 16);
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const int row = it * 4 + trow;
        *reinterpret_cast<uint4 *>(p.out + (int64_t)(mrow0 + row) * p.ldc + ncol0 + tch * 8) = *reinterpret_cast<const uint4 *>(wbuf + row * EROW + tch * 16);
    }
}

// true if it took the launch: plain matrix product, whole tiles, bf16 output, epilogue = bias / GELU or ReLU / residual
bool try_launch_gemm_big(const IgemmParams &p, bool pre, hipStream_t st, float *ws, int64_t ws_floats)
{
    constexpr int min_rows = 4096;        // fixed by measurement (profiles/NOTES.md) (0x7fffffff: off)
    const int64_t M = (int64_t)p.nbatch * p.T_out;
    if (pre || p.taps != 1 || p.stride != 1 || p.pad != 0 || p.T_out != p.T_in || M < min_rows || M % GB_BM || p.N % GB_BN || p.K % GB_BK ||
        p.K < 2 * GB_BK)
        return false;
    if (p.out_f32 || p.colmask || p.accumulate || p.n_split || p.dyn || p.aln_stats || p.rln_stats || p.stats_out || p.zt_cout ||
        p.out_scale != 1.0f || p.ostride != 1 || p.ooff != 0 ||
        !(p.act == ACT_NONE || p.act == ACT_GELU || p.act == ACT_RELU || p.act == ACT_SILU_GLU) || (p.act == ACT_SILU_GLU && p.resid))
        return false;
    // dense [nbatch * T][..] views only: rows of consecutive batch entries are ld apart like the rows inside one
    if (p.x_bstride != (int64_t)p.T_in * p.lda || p.out_bstride != (int64_t)p.T_out * p.ldc ||
        (p.resid && p.resid_bstride != (int64_t)p.T_out * p.resid_ld))
        return false;
    if ((((uintptr_t)p.x) & 15) || (((uintptr_t)p.w) & 15) || (((uintptr_t)p.out) & 15) || (p.resid && (((uintptr_t)p.resid) & 15)) ||
        (p.bias && (((uintptr_t)p.bias) & 15)) || p.lda % 8 || p.ldc % 8 || (p.resid && p.resid_ld % 8))
        return false;
    if ((int64_t)M * p.lda * 2 >= (1ll << 32) || (int64_t)p.N * p.K * 2 >= (1ll << 32)) return false;      // 32-bit DMA offsets
    if (try_launch_gemm_big8(p, M, st, ws, ws_floats)) return true;         // whole 256 x 256 tiles: the eight-wave persistent form (gemm_big8.hip)
    GemmBigParams g;
    g.x = p.x; g.lda = p.lda; g.w = p.w; g.bias = p.bias; g.resid = p.resid; g.ldr = p.resid_ld;
    g.out = (uint16_t *)p.out; g.ldc = p.ldc; g.M = (int)M; g.N = p.N; g.K = p.K;
    g.mtiles = (int)(M / GB_BM); g.ntiles = p.N / GB_BN;
#ifdef GB_DEV_ABL          /* tools builds only: ablations chosen by IFH_GEMM_BIG_ABL (wrong results) */
    g.abl = getenv("IFH_GEMM_BIG_ABL") ? atoi(getenv("IFH_GEMM_BIG_ABL")) : 0;
#else
    g.abl = 0;
#endif
    // IFH_GEMM_BIG_LDS (tuning switch): request that many bytes of LDS instead -- above 80 KB a CU holds ONE workgroup of this kernel
    // and half of its registers stay free for the decode chains' workgroups
    constexpr size_t lds_req = 0;
    const size_t bytes = lds_req > (size_t)GB_STAGES * GB_STAGE_BYTES ? lds_req : (size_t)GB_STAGES * GB_STAGE_BYTES;
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_gemm_big<ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_gemm_big<ACT_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_gemm_big<ACT_RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_gemm_big<ACT_SILU_GLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return false;
        attr_once.done(attr_dev);
    }
    const dim3 grid((unsigned)(g.mtiles * g.ntiles));
    if (p.act == ACT_GELU) hipLaunchKernelGGL(k_gemm_big<ACT_GELU>, grid, dim3(256), bytes, st, g);
    else if (p.act == ACT_RELU) hipLaunchKernelGGL(k_gemm_big<ACT_RELU>, grid, dim3(256), bytes, st, g);
    else if (p.act == ACT_SILU_GLU) hipLaunchKernelGGL(k_gemm_big<ACT_SILU_GLU>, grid, dim3(256), bytes, st, g);
    else hipLaunchKernelGGL(k_gemm_big<ACT_NONE>, grid, dim3(256), bytes, st, g);
    return true;
}

}  // namespace ifh
