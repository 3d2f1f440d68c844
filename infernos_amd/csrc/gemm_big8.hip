// gemm_big8.hip -- the 256 x 256 form of gemm_big.hip: out[M][N] = act(x[M][K] @ w[N][K]^T + bias) (+ resid) for the matrix
// products with thousands of rows (Cluster/InfernSTTWorker.py:65 -> the Whisper encoder layers at 128 windows x 1500 positions,
// HelloSippyRTPipe.py:47-110's text encoder, the LLM's prefill: Cluster/InfernLLMWorker.py:103-119).
//
// k_gemm_big (two workgroups of four waves per CU, 256 x 128 tiles, two 24 KB stages in flight each) spent 41 % of its wave time
// waiting for DMA pieces (profiles/r04_gemm_big_pmc.md): at 85 FLOP per operand byte a CU running the matrix pipe at full rate
// asks L2 for 48 bytes per clock, and 96 KB in flight cover 2 000 clocks of a 3-4 000 clock round trip.  LDS bounds what can be in
// flight, so the lever is the tile: here ONE persistent workgroup of EIGHT waves per CU owns 256 x 256 tiles (128 FLOP per byte:
// 32 bytes per clock at full rate) and a ring of four 32 KB stages, three of them in flight (96 KB = 3 000 clocks).
//   * a stage is 32 blocks of 16 rows x 64 B (one MFMA fragment each; 16 of w, 16 of x), fetched by global_load_lds with the
//     swizzle on the source side as in k_gemm_big; a wave computes 128 columns x 64 rows = 32 accumulator tiles;
//   * fragment reads run one HALF stage ahead of the MFMAs in a second register set (fa_hi under the first 16 MFMAs of a stage, the
//     next stage's fb | fa_lo under the second 16), so the eight waves -- which the one barrier per stage keeps in step -- do not
//     all read, then all multiply;
//   * the workgroup walks its tiles (XCD-contiguous, column tile fastest) and starts the next tile's first three stages from the
//     last iterations of the current one; the epilogue transposes through the fourth ring slot (4 KB per wave, XOR-swizzled
//     256-byte rows, no barrier: a wave only touches its own part) and leaves as whole 256-byte row pieces while those stages land;
//   * vmcnt waits are counted: LDS-DMA, loads and stores retire in issue order (MI355X_MICROARCH.md, s_waitcnt), so the epilogue's
//     stores are part of the count at the next tile's entry.
// Per output element the same ascending chain of v_mfma_f32_16x16x32_bf16 steps and the same epilogue arithmetic as k_igemm /
// k_gemm_big: the same bits (tests/test_nn_gpu.py::test_gemm_big_matches_torch_and_the_igemm_bits).
#include <stdio.h>
#include <stdlib.h>

#include "igemm.h"

namespace ifh {

struct GemmBig8Params {
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
    int stagger;                 // phase-shifted workgroup starts (more than one round of tiles)
    long long *prof;             // tools builds: [0] K-loop clocks, [1] epilogue, [2] entry barrier, [3] tiles (wave 0 of every workgroup)
    int abl;                     // tools builds (GB_DEV_ABL, wrong results): 1 no DMA after the first three stages, 2 no MFMAs, 4 no epilogue
};

constexpr int G8_BN = 256, G8_BM = 256, G8_BK = 32, G8_SLOTS = 4;
constexpr int G8_STAGE = 32 * 1024;                        // 16 + 16 blocks of 1 KB
constexpr int G8_RING = G8_SLOTS * G8_STAGE;
constexpr int G8_EPI = 3 * G8_STAGE;                       // the epilogue's transposition area: ring slot 3
constexpr int G8_BIAS = G8_RING;                           // bias of the tile's columns: [tile parity][wave][128] f32
constexpr int G8_LDS = G8_RING + 2 * 8 * 512;

template <int N>
__device__ __forceinline__ void g8_wait_vm()
{
    static_assert(N == 0 || N == 4 || N == 8 || N == 10 || N == 16 || N == 24, "vmcnt count");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}

template <int ACT, bool RESID>
__global__ __launch_bounds__(512, 2) void k_gemm_big8(const GemmBig8Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef GB_DEV_ABL
    const int abl = p.abl;
#else
    constexpr int abl = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid & 1, wm = wid >> 1;             // wave: columns [128 wn, +128) x rows [64 wm, +64) of the tile
    const int fr = lane & 15, fg = lane >> 4;

    // tiles of this workgroup: round k takes tile k G + (b % 8) (G / 8) + b / 8 -- the workgroups of one XCD (equal b % 8) hold a
    // contiguous run of tiles, column tile fastest: the column tiles of the same rows of x run side by side on one L2
    const int total = p.mtiles * p.ntiles, G = gridDim.x;
    const int nst = p.K / G8_BK;                       // a multiple of 4, >= 8: a tile's stage s lives in ring slot s % 4
    int t = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    if (t >= total) return;

    // DMA: block j of a stage (j < 16: rows n0 + 16 j .. of w; else rows m0 + 16 (j - 16) .. of x) is one wave-instruction: lane l
    // fetches 16 bytes of row l / 4, 16-byte chunk (l % 4) ^ swz(l / 4).  This wave's blocks: wid, wid + 8 (w), 16 + wid, 24 + wid (x).
    // Per-lane parts of the source offsets in two registers; the tile, wave and stage parts are scalar and go into the base pointer.
    const int drow = lane >> 2, dch = (lane & 3) ^ (((drow >> 2) & 1) << 1);
    const unsigned lw = (unsigned)((drow * p.K + dch * 8) * 2), lx = (unsigned)((drow * p.lda + dch * 8) * 2);
    const int64_t whalf = (int64_t)128 * p.K * 2, xhalf = (int64_t)128 * p.lda * 2;
    struct Src { const unsigned char *w, *x, *b; };    // scalar: first w row / x row / bias value of this wave in a tile
    auto tile_src = [&](int tile) {
        const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
        Src r;
        r.w = reinterpret_cast<const unsigned char *>(p.w) + (int64_t)(nt * G8_BN + wid * 16) * p.K * 2;
        r.x = reinterpret_cast<const unsigned char *>(p.x) + (int64_t)(mt * G8_BM + wid * 16) * p.lda * 2;
        // no bias: the pieces are fetched all the same (from w, never read) so that the counted waits do not depend on it
        r.b = p.bias ? reinterpret_cast<const unsigned char *>(p.bias + nt * G8_BN + wn * 128) : reinterpret_cast<const unsigned char *>(p.w);
        return r;
    };
#define G8_DMA(OP, VOFF, BASE, DST)                                                                                       \
    do {                                                                                                                  \
        unsigned keep_;                                                                                                   \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" OP " %1, %2\n\ts_mov_b32 m0, %0"               \
                     : "=&s"(keep_) : "v"(VOFF), "s"(BASE), "s"(DST) : "memory");                                        \
    } while (0)
    // piece q of a stage: q = 0, 1 this wave's two blocks of w, q = 2, 3 of x
    auto issue_piece = [&](const Src &src, int stage_k, int slot, int q) {     // K range [32 stage_k, +32) of a tile into a ring slot
        const int dst = slot * G8_STAGE + wid * 1024 + q * 8 * 1024;
        if (q < 2) {
            const unsigned char *wb = src.w + (int64_t)stage_k * (G8_BK * 2) + (q & 1) * whalf;
            G8_DMA("global_load_lds_dwordx4", lw, wb, dst);
        } else {
            const unsigned char *xb = src.x + (int64_t)stage_k * (G8_BK * 2) + (q & 1) * xhalf;
            G8_DMA("global_load_lds_dwordx4", lx, xb, dst);
        }
    };
    auto issue_stage = [&](const Src &src, int stage_k, int slot) {
#pragma unroll
        for (int q = 0; q < 4; q++) issue_piece(src, stage_k, slot, q);
    };
    // the bias of a tile's columns comes by DMA too (a wave's 128 values = two dword pieces, issued in front of the tile's stage 0):
    // a register load in the epilogue would have to wait for every older operation, i.e. for the next tile's stages
    auto issue_bias = [&](const Src &src, int par) {
        const int dst = G8_BIAS + par * 4096 + wid * 512;
        unsigned lb;                                       // 4 x lane, made here: one register less through the K loop
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0" : "=v"(lb));
        G8_DMA("global_load_lds_dword", lb, src.b, dst);
        G8_DMA("global_load_lds_dword", lb, src.b + 256, dst + 256);
    };
    constexpr int NSTORE = ACT == ACT_SILU_GLU ? 8 : 16;       // global stores of one epilogue, per wave

    // fragment read offset of this lane inside a block: row fr, chunk fg ^ swz(fr)
    const int foff = fr * 64 + ((fg ^ (((fr >> 2) & 1) << 1)) << 4);
    const unsigned char *abase = lds + (wn * 8) * 1024 + foff, *bbase = lds + (16 + wm * 4) * 1024 + foff;

    Src cur = tile_src(t), nxt = cur;
    issue_bias(cur, 0);
    issue_stage(cur, 0, 0);
    issue_stage(cur, 1, 1);
    issue_stage(cur, 2, 2);
    // Every tile takes the same time, so without this all the workgroups would reach their epilogues together: 128 KB of stores per
    // CU = 33 MB at once, as long at the HBM write rate as half a K = 512 loop, with the write path idle in between.  The workgroups
    // start in eight phases an eighth of a tile apart (a tile: ~1 100 clocks per stage + the epilogue).
    if (p.stagger) {
        const int phase = (blockIdx.x >> 3) & 7;
        const int naps = phase * (nst * 1100 + 4000) / (8 * 64 * 100);
        for (int i = 0; i < naps; i++) __builtin_amdgcn_s_sleep(100);
    }
    bool first_tile = true;
    int par = 0;
#ifdef GB_DEV_ABL
    long long pf_k = 0, pf_e = 0, pf_b = 0, pf_n = 0, pf_t = 0;
#define G8_STAMP(ACC) do { if (abl & 8) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); ACC += now_ - pf_t; pf_t = now_; } } while (0)
#else
#define G8_STAMP(ACC) do { } while (0)
#endif

    f32x4 acc[8][4];
    bf16x8_t fb[4], fl[4], fh[4];

    // One stage (ring slot U).  First half: fa_hi of this stage is read under the MFMAs of fa_lo.  Middle: WAIT retires the DMA pieces
    // of the next stage (of this tile, or stage 0 of the next one), one barrier publishes them and says that nobody reads this
    // stage's slot again, ISSUE sends the stage four ahead into it.  Second half: the next stage's fb | fa_lo are read under the
    // MFMAs of fa_hi (READ_NEXT false: a tile's last stage).
#define G8_STAGE_BODY(U, WAIT, ISSUE, READ_NEXT)                                                                          \
    do {                                                                                                                  \
        const unsigned char *cur_ = abase + (U) * G8_STAGE;                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; i++) fh[i] = *reinterpret_cast<const bf16x8_t *>(cur_ + (4 + i) * 1024); \
        if (!(abl & 2)) {                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                 \
                _Pragma("unroll") for (int j = 0; j < 4; j++)                                                             \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[i], fb[j], acc[i][j], 0, 0, 0);                \
            _Pragma("unroll") for (int g = 0; g < 4; g++) {                                                               \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                        \
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                        \
            }                                                                                                             \
        } else {                                                                                                          \
            asm volatile("" ::"v"(fl[0]), "v"(fl[3]), "v"(fb[0]), "v"(fb[3]));                                            \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        if (abl & 1) g8_wait_vm<0>();                                                                                     \
        else { WAIT; }                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
        __builtin_amdgcn_s_barrier();                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        bf16x8_t nb_[4], nl_[4];                                                                                          \
        const unsigned char *nxa_ = abase + (((U) + 1) & 3) * G8_STAGE, *nxb_ = bbase + (((U) + 1) & 3) * G8_STAGE;       \
        /* four groups: one DMA piece, two fragment reads, four MFMAs -- the pieces' issue cost (60-185 clocks each, more  \
           when eight waves issue theirs at once) lies under the other wave's MFMAs instead of stopping both */           \
        _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                                   \
            if (!(abl & 1)) { ISSUE; }                                                                                    \
            if (READ_NEXT) {                                                                                              \
                /* fb first: the next stage's first MFMA wants all of it and fa_lo[0] */                               \
                if (q < 2) {                                                                                              \
                    nb_[2 * q] = *reinterpret_cast<const bf16x8_t *>(nxb_ + (2 * q) * 1024);                              \
                    nb_[2 * q + 1] = *reinterpret_cast<const bf16x8_t *>(nxb_ + (2 * q + 1) * 1024);                      \
                } else {                                                                                                  \
                    nl_[2 * q - 4] = *reinterpret_cast<const bf16x8_t *>(nxa_ + (2 * q - 4) * 1024);                      \
                    nl_[2 * q - 3] = *reinterpret_cast<const bf16x8_t *>(nxa_ + (2 * q - 3) * 1024);                      \
                }                                                                                                         \
            }                                                                                                             \
            if (!(abl & 2)) {                                                                                             \
                _Pragma("unroll") for (int j = 0; j < 4; j++)                                                             \
                    acc[4 + q][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[q], fb[j], acc[4 + q][j], 0, 0, 0);        \
                if (READ_NEXT) {                                                                                          \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                    \
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                    \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                    \
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                    \
                }                                                                                                         \
            } else {                                                                                                      \
                asm volatile("" ::"v"(fh[q]), "v"(fb[0]), "v"(fb[3]));                                                    \
            }                                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
        }                                                                                                                 \
        if (READ_NEXT) {                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; j++) { fb[j] = nb_[j]; fl[j] = nl_[j]; }                             \
        }                                                                                                                 \
    } while (0)

    for (;;) {
        const int tnext = t + G;
        const bool has_next = tnext < total;
        if (has_next) nxt = tile_src(tnext);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- entry.  Stage 0 was published by the previous tile's last barrier (first tile: wait for it here; younger: stages 1, 2).
        // The barrier says every wave has left the epilogue's part of slot 3: stage 3 goes into it.
        if (first_tile) g8_wait_vm<8>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef GB_DEV_ABL
        if (first_tile) pf_t = (long long)__builtin_amdgcn_s_memtime();
        else G8_STAMP(pf_b);
        pf_n++;
#endif
        if (!(abl & 1) || first_tile) issue_stage(cur, 3, 3);
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j] = *reinterpret_cast<const bf16x8_t *>(bbase + j * 1024);
#pragma unroll
        for (int i = 0; i < 4; i++) fl[i] = *reinterpret_cast<const bf16x8_t *>(abase + i * 1024);

        // ---- all but the last four stages.  Younger than the awaited stage s + 1: the two stages behind it, and for s < 2 the previous
        // tile's epilogue stores (issued between its prefetch of stage 2 and this tile's stage 3)
        // (the group that follows an epilogue is marked by an opaque value: from a plain flag hipcc peels it off the loop, and the second
        // copy of the stage code costs registers -- half an accumulator tile went to scratch, i.e. into the DMA queue)
        int s_epi = __builtin_amdgcn_readfirstlane(first_tile ? -1 : 0);
        asm volatile("" : "+s"(s_epi));
        first_tile = false;
        int s0 = 0;
        do {
            const bool after_epi = s0 == s_epi;
            G8_STAGE_BODY(0, if (after_epi) g8_wait_vm<8 + NSTORE>(); else g8_wait_vm<8>(), issue_piece(cur, s0 + 4, 0, q), true);
            G8_STAGE_BODY(1, if (after_epi) g8_wait_vm<8 + NSTORE>(); else g8_wait_vm<8>(), issue_piece(cur, s0 + 5, 1, q), true);
            G8_STAGE_BODY(2, g8_wait_vm<8>(), issue_piece(cur, s0 + 6, 2, q), true);
            G8_STAGE_BODY(3, g8_wait_vm<8>(), issue_piece(cur, s0 + 7, 3, q), true);
            s0 += 4;
        } while (s0 < nst - 4);
        // ---- the last four stages start the next tile: its bias pieces and stage 0, then stages 1 and 2 (stage 3 waits for the
        // epilogue to leave slot 3).  Without a next tile the ring drains.
        G8_STAGE_BODY(0, g8_wait_vm<8>(), if (has_next) { if (q == 0) issue_bias(nxt, par ^ 1); issue_piece(nxt, 0, 0, q); }, true);
        G8_STAGE_BODY(1, if (has_next) g8_wait_vm<10>(); else g8_wait_vm<4>(), if (has_next) issue_piece(nxt, 1, 1, q), true);
        G8_STAGE_BODY(2, if (has_next) g8_wait_vm<10>(); else g8_wait_vm<0>(), if (has_next) issue_piece(nxt, 2, 2, q), true);
        G8_STAGE_BODY(3, if (has_next) g8_wait_vm<8>(), (void)0, false);

        // ---- epilogue.  D[n][m]: a lane holds 4 consecutive columns n of row m = fr of accumulator tile (i, j).  Piece j = 16 rows x 128
        // columns of the wave's sub-tile goes through the wave's 4 KB of slot 3 (16 rows x 256 B, 16-byte chunk c of row r at chunk
        // c ^ r) and leaves as 256-byte row pieces, 16 bytes per lane; the residual comes in the same way and is added in f32 before
        // the one rounding, as k_igemm does.
        G8_STAMP(pf_k);
        if (!(abl & 4)) {
            // the epilogue's addresses are functions of the lane number only: taken from an opaque copy, or hipcc computes them once in
            // front of the tile loop and carries ~20 more registers through the K loop (spills, whose reloads sit in the DMA queue)
            int el, te = t;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));       // the lane number
            asm volatile("" : "+s"(te));
            const int mt = te / p.ntiles, nt = te - mt * p.ntiles;
            const int mrow0 = mt * G8_BM + wm * 64, ncol0 = nt * G8_BN + wn * 128;
            const int efr = el & 15, efg = el >> 4, trow = el >> 4, tch = el & 15;
            unsigned char *wbuf = lds + G8_EPI + wid * 4096;
            const unsigned char *bl = lds + G8_BIAS + par * 4096 + wid * 512 + efg * 16;
            const bool hb = p.bias != nullptr;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (RESID) {
                    uint4 rr[4];
#pragma unroll
                    for (int it = 0; it < 4; it++) {
                        const int row = it * 4 + trow;
                        rr[it] = *reinterpret_cast<const uint4 *>(p.resid + (int64_t)(mrow0 + j * 16 + row) * p.ldr + ncol0 + tch * 8);
                    }
#pragma unroll
                    for (int it = 0; it < 4; it++) {
                        const int row = it * 4 + trow;
                        *reinterpret_cast<uint4 *>(wbuf + row * 256 + ((tch ^ row) << 4)) = rr[it];
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
                    if (hb) {
                        const float4 bv = *reinterpret_cast<const float4 *>(bl + i * 64);
                        v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
                    }
                    if (ACT == ACT_SILU_GLU) {
                        // interleaved (gate, up) weight rows: the lane's four columns are two pairs -> two outputs of the half-width
                        // result (k_gemm_m64's arithmetic, nn.hip); 128-byte rows, 16-byte chunk i at chunk i ^ (fr & 7)
                        const float o0 = v0 / (1.0f + __expf(-v0)) * v1, o1 = v2 / (1.0f + __expf(-v2)) * v3;
                        *reinterpret_cast<uint32_t *>(wbuf + efr * 128 + ((i ^ (efr & 7)) << 4) + efg * 4) = f32x2_to_bf16x2(o0, o1);
                        continue;
                    }
                    if (ACT != ACT_NONE) {
                        v0 = apply_act_c<ACT>(v0, ACT, 0.0f); v1 = apply_act_c<ACT>(v1, ACT, 0.0f);
                        v2 = apply_act_c<ACT>(v2, ACT, 0.0f); v3 = apply_act_c<ACT>(v3, ACT, 0.0f);
                    }
                    unsigned char *slot = wbuf + efr * 256 + (((i * 2 + (efg >> 1)) ^ efr) << 4) + (efg & 1) * 8;
                    if (RESID) {
                        const uint2 rv = *reinterpret_cast<const uint2 *>(slot);
                        v0 += __uint_as_float(rv.x << 16);
                        v1 += __uint_as_float(rv.x & 0xffff0000u);
                        v2 += __uint_as_float(rv.y << 16);
                        v3 += __uint_as_float(rv.y & 0xffff0000u);
                    }
                    *reinterpret_cast<uint2 *>(slot) = make_uint2(f32x2_to_bf16x2(v0, v1), f32x2_to_bf16x2(v2, v3));
                }
                if (ACT == ACT_SILU_GLU) {                 // 64 outputs per row: 128 bytes, eight 16-byte pieces; 8 rows per instruction
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const int row = it * 8 + (el >> 3), ch = el & 7;
                        *reinterpret_cast<uint4 *>(p.out + (int64_t)(mrow0 + j * 16 + row) * p.ldc + (ncol0 >> 1) + ch * 8) =
                            *reinterpret_cast<const uint4 *>(wbuf + row * 128 + ((ch ^ (row & 7)) << 4));
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 4; it++) {
                        const int row = it * 4 + trow;
                        *reinterpret_cast<uint4 *>(p.out + (int64_t)(mrow0 + j * 16 + row) * p.ldc + ncol0 + tch * 8) =
                            *reinterpret_cast<const uint4 *>(wbuf + row * 256 + ((tch ^ row) << 4));
                    }
                }
            }
        }
        G8_STAMP(pf_e);
        if (!has_next) break;
        t = tnext;
        cur = nxt;
        par ^= 1;
    }
#ifdef GB_DEV_ABL
    if ((abl & 8) && p.prof && tid == 0) {
        atomicAdd((unsigned long long *)p.prof + 0, (unsigned long long)pf_k);
        atomicAdd((unsigned long long *)p.prof + 1, (unsigned long long)pf_e);
        atomicAdd((unsigned long long *)p.prof + 2, (unsigned long long)pf_b);
        atomicAdd((unsigned long long *)p.prof + 3, (unsigned long long)pf_n);
    }
#endif
#undef G8_STAGE_BODY
#undef G8_DMA
#undef G8_STAMP
}

// true if it took the launch (try_launch_gemm_big has checked the epilogue and the views; here: whole 256 x 256 tiles, K in 128s)
bool try_launch_gemm_big8(const IgemmParams &p, int64_t M, hipStream_t st)
{
    static const int on = getenv("IFH_GEMM_BIG8") ? atoi(getenv("IFH_GEMM_BIG8")) : 1;      // tuning switch: 0 = the 256 x 128 kernel
    if (!on || M % G8_BM || p.N % G8_BN || p.K % (4 * G8_BK) || p.K < 8 * G8_BK) return false;
    GemmBig8Params g;
    g.x = p.x; g.lda = p.lda; g.w = p.w; g.bias = p.bias; g.resid = p.resid; g.ldr = p.resid_ld;
    g.out = (uint16_t *)p.out; g.ldc = p.ldc; g.M = (int)M; g.N = p.N; g.K = p.K;
    g.mtiles = (int)(M / G8_BM); g.ntiles = p.N / G8_BN;
    static const int stagger = getenv("IFH_GEMM_BIG8_STAGGER") ? atoi(getenv("IFH_GEMM_BIG8_STAGGER")) : 1;       // tuning switch
    g.stagger = stagger && g.mtiles * g.ntiles > 2 * device_cu_count();
    g.prof = nullptr;
#ifdef GB_DEV_ABL          /* tools builds only: ablations chosen by IFH_GEMM_BIG_ABL (wrong results; 8 = phase clocks, printed per launch) */
    g.abl = getenv("IFH_GEMM_BIG_ABL") ? atoi(getenv("IFH_GEMM_BIG_ABL")) : 0;
    static long long *prof_buf = nullptr;
    if (g.abl & 8) {
        if (!prof_buf && hipMalloc((void **)&prof_buf, 64) != hipSuccess) return false;
        (void)hipMemsetAsync(prof_buf, 0, 64, st);
        g.prof = prof_buf;
    }
#else
    g.abl = 0;
#endif
    const void *fn;
    if (p.act == ACT_GELU) fn = p.resid ? (const void *)k_gemm_big8<ACT_GELU, true> : (const void *)k_gemm_big8<ACT_GELU, false>;
    else if (p.act == ACT_RELU) fn = p.resid ? (const void *)k_gemm_big8<ACT_RELU, true> : (const void *)k_gemm_big8<ACT_RELU, false>;
    else if (p.act == ACT_SILU_GLU) fn = (const void *)k_gemm_big8<ACT_SILU_GLU, false>;
    else fn = p.resid ? (const void *)k_gemm_big8<ACT_NONE, true> : (const void *)k_gemm_big8<ACT_NONE, false>;
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        const void *all[] = {(const void *)k_gemm_big8<ACT_GELU, true>, (const void *)k_gemm_big8<ACT_GELU, false>,
                             (const void *)k_gemm_big8<ACT_RELU, true>, (const void *)k_gemm_big8<ACT_RELU, false>,
                             (const void *)k_gemm_big8<ACT_SILU_GLU, false>,
                             (const void *)k_gemm_big8<ACT_NONE, true>, (const void *)k_gemm_big8<ACT_NONE, false>};
        for (const void *f : all)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS) != hipSuccess) return false;
        attr_once.done(attr_dev);
    }
    const int total = g.mtiles * g.ntiles;
    int grid = device_cu_count() & ~7;
    if (grid < 8) grid = 8;
    if (grid > total) grid = total < 8 ? total : (total & ~7);
    // with fewer than 8 tiles the XCD interleave below degenerates: one workgroup per tile
    void *args[] = {(void *)&g};
    if (total < 8) {
        // first = (b & 7) * (G >> 3) + (b >> 3) needs G >= 8; run as 8 workgroups, the surplus ones return at once
        grid = 8;
    }
    const bool ok = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(512), args, G8_LDS, st) == hipSuccess;
#ifdef GB_DEV_ABL
    if (ok && g.prof) {
        long long h[4] = {0, 0, 0, 0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, g.prof, sizeof h, hipMemcpyDeviceToHost);
        if (h[3])
            fprintf(stderr, "k_gemm_big8 %d x %d x %d: per tile (wave 0, shader clocks): K loop %.1f  epilogue %.1f  entry barrier %.1f   (%lld tiles, grid %d)\n",
                    g.M, g.N, g.K, (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3], h[3], grid);
    }
#endif
    return ok;
}

}  // namespace ifh
