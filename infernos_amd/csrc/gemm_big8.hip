// gemm_big8.hip -- the 256 x 256 form of gemm_big.hip: out[M][N] = act(x[M][K] @ w[N][K]^T + bias) (+ resid) for the matrix
// products with thousands of rows (Cluster/InfernSTTWorker.py:65 -> the Whisper encoder layers at 128 windows x 1500 positions,
// HelloSippyRTPipe.py:47-110's text encoder, the LLM's prefill: Cluster/InfernLLMWorker.py:103-119).
//
// k_gemm_big (two workgroups of four waves per CU, 256 x 128 tiles, 16-row x 64-byte DMA pieces) spent 41 % of its wave time waiting
// for DMA pieces (profiles/r04_gemm_big_pmc.md); the first eight-wave form of this file (256 x 256 tiles, four 32 KB stages of the same
// piece shape) ran the K loop at 1 850 clocks per 32 of K against 1 024 of MFMAs with the texture-address unit busy 64 % of the time:
// a 16-row x 64-byte piece touches sixteen 128-byte lines for half of each.  Here
//   * ONE persistent workgroup of EIGHT waves per CU owns 256 x 256 tiles (128 FLOP per operand byte); a wave computes 128 columns x
//     64 rows = 32 accumulator tiles;
//   * a stage is 64 of K: 64 units of 8 rows x 128 B (32 of w, 32 of x), each one global_load_lds wave-instruction fetching whole
//     lines; inside a unit the 16-byte chunk c of row r lies at chunk c ^ 2 (r / 2) (the permutation is applied on the source side, the
//     DMA writes LDS linearly), which makes the ds_read_b128 fragment reads of both 32-wide k steps conflict-free;
//   * two stages of 64 KB are all LDS holds, so the ring is two deep: a stage's slot is refilled with the stage two ahead as soon as
//     every wave has read its last fragments (the one barrier per stage, three quarters through it), four pieces under the last
//     sixteen MFMAs of the stage and four under the first sixteen of the next;
//   * fragment reads run one quarter stage (16 MFMAs) ahead of the MFMAs in a second register set, so the eight waves -- which the
//     barrier keeps in step -- do not all read, then all multiply;
//   * the workgroup walks its tiles (XCD-contiguous, column tile fastest) and fetches the next tile's first stage under the current
//     one's last two; a wave's DMA units are also its epilogue buffers (the slot-1 halves of its four w units and of its four x units:
//     two 4 KB buffers of XOR-swizzled 256-byte rows), so the epilogue needs no barrier and the next tile's stage 1 goes out as each
//     wave leaves it; outputs leave as whole 256-byte row pieces; the residual rows arrive by DMA in the same shape, one piece ahead
//     (the first two under the tile's last sixteen MFMAs); the tile's bias comes by DMA as well (a register load in the epilogue
//     would have to wait for every older operation in the queue, i.e. for the next tile's stage).
//   * with a workspace of the caller's the tiles of the last, partial round (288 tiles on 256 CUs: the LLM prompt) are cut into parts of
//     K whose raw f32 accumulators k_big8_split_finish adds in order before the epilogue; the workgroup count follows the process-wide
//     CU budget unless the launch says it runs alone (ifh_conv_desc.whole_chip).
// Measured (192 000 rows, tools/probe_igemm_enc.py, profiles/r05_gemm_big_pmc.md): q|k|v 1536 x 512 830 TF/s (k_gemm_big 610), fc2
// 512 x 2048 1 075-1 165, fc1 + GELU 680-700, wo + residual 680.  Per tile of the q|k|v product wave 0 spends 30 k clocks in the K loop
// (16.4 k of MFMAs; 1 100 per stage in front of and inside the stage barrier: the 64 KB of a stage arrive at ~40 B/clk, the L2 -> LDS
// rate of a CU with every CU fetching, and two slots cannot keep the fetch running through the barrier), 4 k in the epilogue, and the
// stores of all workgroups fall together (without them 300 us instead of 380: started in phases it did not change).
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
    float *split_ws;             // null, or: the tiles of the last (partial) round are cut into split_s parts of K, each part's raw f32
    int split_s;                 // accumulators go to split_ws[tile - first leftover tile][part][256 rows][256 columns]; k_big8_split_finish follows
    long long *prof;             // tools builds: [0] K-loop clocks, [1] epilogue, [2] entry barrier, [3] tiles (wave 0 of every workgroup)
    int abl;                     // tools builds (GB_DEV_ABL, wrong results): 1 no DMA after a tile's first stage, 2 no MFMAs, 4 no epilogue, 16 no output stores
};

constexpr int G8_BN = 256, G8_BM = 256, G8_DK = 64;
constexpr int G8_RING = 2 * 64 * 1024;                     // unit u of slot sl at u * 2048 + sl * 1024
constexpr int G8_BIAS = G8_RING;                           // bias of the tile's columns: [tile parity][wave][128] f32
constexpr int G8_LDS = G8_RING + 2 * 8 * 512;

template <int N>
__device__ __forceinline__ void g8_wait_vm()
{
    static_assert(N == 0 || N == 8 || N == 16, "vmcnt count");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
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
    const int nds_full = p.K / G8_DK;                  // stages per tile: even, >= 4; stage s lives in slot s % 2
    // The workgroup's items: tile first + k G of every full round k; then its item of the last, partial round -- a whole tile, or
    // (split_ws) part j % S of tile R G + j / S, j = first: with 288 tiles on 256 CUs the 32 tiles of the second round would otherwise
    // run alone on 32 CUs for a whole tile time.  A part is nds_full / S stages (even, >= 4) and leaves raw f32 accumulators.
    const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int R = total / G, S = p.split_ws ? p.split_s : 1, nds_part = nds_full / S;
    // item k of this workgroup -> tile (or -1) and part (-1: the whole K range)
    auto item_tile = [&](int k) { return k < R ? first + k * G : (k > R ? -1 : (S > 1 ? (first < (total - R * G) * S ? R * G + first / S : -1)
                                                                                  : (first + R * G < total ? first + R * G : -1))); };
    auto item_part = [&](int k) { return (k == R && S > 1) ? first % S : -1; };
    int kitem = 0;
    int t = item_tile(0), tpart = item_part(0);
    if (t < 0) return;
    int nds = tpart < 0 ? nds_full : nds_part;

    // DMA: unit u of a stage (u < 32: rows n0 + 8 u .. of w; else rows m0 + 8 (u - 32) .. of x) is one wave-instruction: lane l fetches
    // the 16 bytes that belong at LDS position (row l / 8, chunk l % 8), i.e. source chunk (l % 8) ^ 2 (row / 2).  This wave's units:
    // 4 wid + q (w) and 32 + 4 wid + q (x), q = 0..3 -- the slot-1 halves of exactly these eight units are its epilogue buffers, so a wave
    // that has left its epilogue may refill them without asking the others.  Per-lane parts of the source offsets in two registers; tile, wave, unit and stage
    // parts are scalar and go into the base pointer.
    const int drow = lane >> 3, dch = (lane & 7) ^ ((drow >> 1) << 1);
    const unsigned lw = (unsigned)((drow * p.K) * 2 + dch * 16), lx = (unsigned)((drow * p.lda) * 2 + dch * 16);
    const int64_t wq = (int64_t)8 * p.K * 2, xq = (int64_t)8 * p.lda * 2;        // one unit on = 8 rows on
    struct Src { const unsigned char *w, *x, *b; };    // scalar: first w row / x row / bias value of this wave in a tile
    auto tile_src = [&](int tile, int part) {
        const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
        const int64_t koff = part > 0 ? (int64_t)part * nds_part * (G8_DK * 2) : 0;      // a part starts at its own K offset
        Src r;
        r.w = reinterpret_cast<const unsigned char *>(p.w) + (int64_t)(nt * G8_BN + wid * 32) * p.K * 2 + koff;
        r.x = reinterpret_cast<const unsigned char *>(p.x) + (int64_t)(mt * G8_BM + wid * 32) * p.lda * 2 + koff;
        // no bias: the pieces are fetched all the same (from w, never read) -- one instruction stream for both cases
        r.b = p.bias ? reinterpret_cast<const unsigned char *>(p.bias + nt * G8_BN + wn * 128) : reinterpret_cast<const unsigned char *>(p.w);
        return r;
    };
#define G8_DMA(OP, VOFF, BASE, DST)                                                                                       \
    do {                                                                                                                  \
        unsigned keep_;                                                                                                   \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" OP " %1, %2\n\ts_mov_b32 m0, %0"               \
                     : "=&s"(keep_) : "v"(VOFF), "s"(BASE), "s"(DST) : "memory");                                        \
    } while (0)
    // piece q of a stage: q = 0..3 this wave's four units of w, q = 4..7 of x
    auto issue_piece = [&](const Src &src, int stage, int slot, int q) {     // K range [64 stage, +64) of a tile into a slot
        if (q < 4) {
            const unsigned char *wb = src.w + (int64_t)stage * (G8_DK * 2) + q * wq;
            G8_DMA("global_load_lds_dwordx4", lw, wb, (4 * wid + q) * 2048 + slot * 1024);
        } else {
            const unsigned char *xb = src.x + (int64_t)stage * (G8_DK * 2) + (q - 4) * xq;
            G8_DMA("global_load_lds_dwordx4", lx, xb, (32 + 4 * wid + (q - 4)) * 2048 + slot * 1024);
        }
    };
    auto issue_bias = [&](const Src &src, int par) {       // a wave's 128 values = two dword pieces
        const int dst = G8_BIAS + par * 4096 + wid * 512;
        unsigned lb;                                       // 4 x lane, made here: one register less through the K loop
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0" : "=v"(lb));
        G8_DMA("global_load_lds_dword", lb, src.b, dst);
        G8_DMA("global_load_lds_dword", lb, src.b + 256, dst + 256);
    };

    // fragment reads: block i of w = units 16 wn + 2 i, + 1 (row fr: unit fr / 8, row fr % 8); k step h: chunks 4 h + fg, stored at
    // chunk ^ 2 (row / 2).  Four address registers (w / x, k step); block and slot are immediates (i * 4096 + slot * 1024).
    const int rr8 = fr & 7, fsw = (rr8 >> 1) << 1;
    const unsigned char *a0 = lds + (16 * wn + (fr >> 3)) * 2048 + rr8 * 128 + ((fg ^ fsw) << 4);
    const unsigned char *a1 = lds + (16 * wn + (fr >> 3)) * 2048 + rr8 * 128 + (((4 + fg) ^ fsw) << 4);
    const unsigned char *b0 = lds + (32 + 8 * wm + (fr >> 3)) * 2048 + rr8 * 128 + ((fg ^ fsw) << 4);
    const unsigned char *b1 = lds + (32 + 8 * wm + (fr >> 3)) * 2048 + rr8 * 128 + (((4 + fg) ^ fsw) << 4);

    Src cur = tile_src(t, tpart), nxt = cur;
    issue_bias(cur, 0);
#pragma unroll
    for (int q = 0; q < 8; q++) issue_piece(cur, 0, 0, q);
    bool first_tile = true;
    int par = 0;
#ifdef GB_DEV_ABL
    long long pf_k = 0, pf_e = 0, pf_b = 0, pf_n = 0, pf_t = 0;
#define G8_STAMP(ACC) do { if (abl & 8) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); ACC += now_ - pf_t; pf_t = now_; } } while (0)
    long long pf_v = 0, pf_s = 0, pf_m = 0;
    // around a stage's barrier: [0] before the DMA wait, [1] after it, [2] after the barrier (the stamps drain the LDS queue where the
    // kernel does so anyway)
#define G8_MID_STAMP(W)                                                                                                  \
    do {                                                                                                                  \
        if (abl & 8) {                                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
            const long long now_ = (long long)__builtin_amdgcn_s_memtime();                                               \
            if ((W) == 1) pf_v += now_ - pf_m;                                                                            \
            if ((W) == 2) pf_s += now_ - pf_m;                                                                            \
            pf_m = now_;                                                                                                  \
        }                                                                                                                 \
    } while (0)
#else
#define G8_STAMP(ACC) do { } while (0)
#define G8_MID_STAMP(W) do { } while (0)
#endif

    f32x4 acc[8][4];
    bf16x8_t fb[4], fl[4], fh[4], gb[4], gl[4];        // k-lo fragments (fb | fa_lo | fa_hi) and the set being read ahead

#define G8_LD(P, OFF) (*reinterpret_cast<const bf16x8_t *>((P) + (OFF)))
    // sixteen MFMAs acc[IB + i][j] += A[i] x B[j] with up to eight fragment reads and four DMA pieces spread among them
#define G8_QUARTER(IB, A, B, READS, DMA)                                                                                  \
    do {                                                                                                                  \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; g_++) {                                                                \
            if (!(abl & 1)) { const int q = g_; (void)q; DMA; }                                                           \
            { const int g = g_; (void)g; READS; }                                                                         \
            if (!(abl & 2)) {                                                                                             \
                _Pragma("unroll") for (int j = 0; j < 4; j++)                                                             \
                    acc[(IB) + g_][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[g_], B[j], acc[(IB) + g_][j], 0, 0, 0); \
            } else {                                                                                                      \
                asm volatile("" ::"v"(A[g_]), "v"(B[0]), "v"(B[3]));                                                      \
            }                                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
        }                                                                                                                 \
    } while (0)
    // One stage in slot SL.  Q1: k-lo, w blocks 0-3 (fl x fb), fa_hi read; Q2: k-lo, blocks 4-7 (fh x fb), the k-hi fb | fa_lo read
    // into (gb | gl); Q3: k-hi, blocks 0-3 (gl x gb), k-hi fa_hi read into fh; the barrier: the next stage has landed (nothing younger
    // is in the queue: vmcnt(0)) and nobody reads this slot again; Q4: k-hi, blocks 4-7 (fh x gb), the next stage's k-lo fb | fa_lo
    // read into (fb | fl) -- fl and fb are free by then.  DMA1: pieces 4-7 of the next stage (other slot), DMA4: pieces 0-3 of the
    // stage after it (this slot).
#define G8_STAGE(SL, WAIT, DMA1, DMA4, READ_NEXT)                                                                               \
    do {                                                                                                                  \
        constexpr int so_ = (SL) * 1024, no_ = (1 - (SL)) * 1024;                                                         \
        G8_QUARTER(0, fl, fb, fh[g] = G8_LD(a0, (4 + g) * 4096 + so_), DMA1);                                             \
        G8_QUARTER(4, fh, fb, (g < 2 ? (gb[2 * g] = G8_LD(b1, (2 * g) * 4096 + so_), gb[2 * g + 1] = G8_LD(b1, (2 * g + 1) * 4096 + so_)) \
                                     : (gl[2 * g - 4] = G8_LD(a1, (2 * g - 4) * 4096 + so_), gl[2 * g - 3] = G8_LD(a1, (2 * g - 3) * 4096 + so_))), (void)0); \
        G8_QUARTER(0, gl, gb, fh[g] = G8_LD(a1, (4 + g) * 4096 + so_), (void)0);                                          \
        G8_MID_STAMP(0);                                                                                                  \
        if (!(abl & 1)) { WAIT; }                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
        G8_MID_STAMP(1);                                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                                     \
        G8_MID_STAMP(2);                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        if (READ_NEXT)                                                                                                    \
            G8_QUARTER(4, fh, gb, (g < 2 ? (fb[2 * g] = G8_LD(b0, (2 * g) * 4096 + no_), fb[2 * g + 1] = G8_LD(b0, (2 * g + 1) * 4096 + no_)) \
                                         : (fl[2 * g - 4] = G8_LD(a0, (2 * g - 4) * 4096 + no_), fl[2 * g - 3] = G8_LD(a0, (2 * g - 3) * 4096 + no_))), DMA4); \
        else                                                                                                              \
            G8_QUARTER(4, fh, gb, (void)0, DMA4);                                                                         \
    } while (0)

    for (;;) {
        const int tnext = item_tile(kitem + 1), pnext = item_part(kitem + 1);
        const bool has_next = tnext >= 0;
        if (has_next) nxt = tile_src(tnext, pnext);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- entry.  Stage 0 was published by the previous tile's last barrier (first tile: wait for it here).  Stage 1 goes into
        // slot 1 (pieces 0-3 here, 4-7 under stage 0's first MFMAs): into this wave's own units, which only its own epilogue used.
        if (first_tile) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#ifdef GB_DEV_ABL
        if (first_tile) pf_t = (long long)__builtin_amdgcn_s_memtime();
        else G8_STAMP(pf_b);
        pf_n++;
#endif
        if (!(abl & 1) || first_tile) {
#pragma unroll
            for (int q = 0; q < 4; q++) issue_piece(cur, 1, 1, q);
        }
        first_tile = false;
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j] = G8_LD(b0, j * 4096);
#pragma unroll
        for (int i = 0; i < 4; i++) fl[i] = G8_LD(a0, i * 4096);

        // ---- all but the last two stages.  A stage's barrier waits for the next stage's pieces: nothing younger is in the queue.
        int s0 = 0;
        do {
            G8_STAGE(0, g8_wait_vm<0>(), issue_piece(cur, s0 + 1, 1, 4 + q), issue_piece(cur, s0 + 2, 0, q), true);
            G8_STAGE(1, g8_wait_vm<0>(), issue_piece(cur, s0 + 2, 0, 4 + q), issue_piece(cur, s0 + 3, 1, q), true);
            s0 += 2;
        } while (s0 < nds - 2);
        // ---- the last two start the next tile: its stage 0 (and bias) into slot 0; slot 1 stays free for the epilogue
        G8_STAGE(0, g8_wait_vm<0>(), issue_piece(cur, s0 + 1, 1, 4 + q), if (has_next) issue_piece(nxt, 0, 0, q), true);
        // (with a residual: the geometry of the epilogue and the residual rows of its first two pieces, started under the tile's last
        // sixteen MFMAs -- slot 1 is free from the stage's barrier on)
        int el = 0, mrow0 = 0, ncol0 = 0;
        unsigned lres[4] = {0, 0, 0, 0};
        auto epi_geometry = [&]() {
            // the epilogue's addresses are functions of the lane number only: taken from an opaque copy, or hipcc computes them once in
            // front of the tile loop and carries ~20 more registers through the K loop (spills, whose reloads sit in the DMA queue)
            int te = t;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));       // the lane number
            asm volatile("" : "+s"(te));
            const int mt = te / p.ntiles, nt = te - mt * p.ntiles;
            mrow0 = mt * G8_BM + wm * 64;
            ncol0 = nt * G8_BN + wn * 128;
            if (RESID) {
                const int trow = el >> 4, tch = el & 15;
#pragma unroll
                for (int it = 0; it < 4; it++) lres[it] = (unsigned)(trow * p.ldr * 2 + ((tch ^ trow ^ (it << 2)) << 4));
            }
        };
        // residual rows 4 it .. 4 it + 3 of piece j -> the piece's buffer (4 rows x 256 B, the chunk swizzle on the source side)
        auto resid_dma = [&](int j, int it) {
            const unsigned char *rb = reinterpret_cast<const unsigned char *>(p.resid) + ((int64_t)(mrow0 + j * 16 + it * 4) * p.ldr + ncol0) * 2;
            G8_DMA("global_load_lds_dwordx4", lres[it], rb, (((j & 1) ? 32 + 4 * wid : 4 * wid) + it) * 2048 + 1024);
        };
        G8_STAGE(1, g8_wait_vm<0>(), if (has_next) { if (q == 0) issue_bias(nxt, par ^ 1); issue_piece(nxt, 0, 0, 4 + q); },
                 if (RESID && tpart < 0) { if (q == 0) epi_geometry(); resid_dma(q >> 1, 2 * (q & 1)); resid_dma(q >> 1, 2 * (q & 1) + 1); }, false);
        G8_STAMP(pf_k);

        // ---- epilogue.  D[n][m]: a lane holds 4 consecutive columns n of row m = fr of accumulator tile (i, j).  Piece j = 16 rows x 128
        // columns of the wave's sub-tile goes through one of the wave's two 4 KB buffers (the slot-1 halves of its w units 4 wid .. + 3,
        // of its x units 32 + 4 wid .. + 3: row r in unit r / 4, 256 B, 16-byte chunk c at chunk c ^ r) and leaves as 256-byte row pieces,
        // 16 bytes per lane; the residual comes in the same way and is added in f32 before the one rounding, as k_igemm does.
        if (tpart >= 0) {
            // a part of a tile of the last round: raw accumulators to the workspace, [row m][column n] f32 (k_big8_split_finish adds the
            // parts in order and runs the epilogue)
            int el2;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el2));
            float *wp = p.split_ws + ((int64_t)(t - R * G) * S + tpart) * 65536 + (wm * 64 + (el2 & 15)) * 256 + wn * 128 + 4 * (el2 >> 4);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 8; i++)
                    *reinterpret_cast<float4 *>(wp + j * 16 * 256 + i * 16) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        } else if (!(abl & 4)) {
            if (!RESID) epi_geometry();
            const int efr = el & 15, efg = el >> 4, trow = el >> 4, tch = el & 15;
            unsigned char *const ebuf0 = lds + (4 * wid) * 2048 + 1024, *const ebuf1 = lds + (32 + 4 * wid) * 2048 + 1024;
            // the bias of the lane's 32 columns, once per tile (the fragment registers are free now)
            const unsigned char *bl = lds + G8_BIAS + par * 4096 + wid * 512 + efg * 16;
            float4 bv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) bv[i] = *reinterpret_cast<const float4 *>(bl + i * 64);
            if (!p.bias) {
#pragma unroll
                for (int i = 0; i < 8; i++) bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // the residual rows of piece j come by DMA into the piece's buffer (four instructions of 4 rows x 256 B, the chunk swizzle on
            // the source side); pieces 0 and 1 were started under the tile's last sixteen MFMAs.  Counted waits: R0 R1 | wait R0 (younger:
            // R1), S0, R2 | wait R1 (S0 R2), S1, R3 | wait R2 (S1 R3), S2 | wait R3 (S2), S3
#pragma unroll
            for (int j = 0; j < 4; j++) {
                unsigned char *wbuf = (j & 1) ? ebuf1 : ebuf0;
                unsigned char *slot[8];
#pragma unroll
                for (int i = 0; i < 8; i++)
                    slot[i] = wbuf + (efr >> 2) * 2048 + (efr & 3) * 256 + (((i * 2 + (efg >> 1)) ^ efr) << 4) + (efg & 1) * 8;
                uint2 rv[8];
                if (RESID) {
                    if (j == 0 || j == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
                    for (int i = 0; i < 8; i++) rv[i] = *reinterpret_cast<const uint2 *>(slot[i]);
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y, v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
                    if (ACT == ACT_SILU_GLU) {
                        // interleaved (gate, up) weight rows: the lane's four columns are two pairs -> two outputs of the half-width
                        // result (k_gemm_m64's arithmetic, nn.hip); 128-byte rows (8 per unit), 16-byte chunk i at chunk i ^ (fr & 7)
                        const float o0 = v0 / (1.0f + __expf(-v0)) * v1, o1 = v2 / (1.0f + __expf(-v2)) * v3;
                        *reinterpret_cast<uint32_t *>(wbuf + (efr >> 3) * 2048 + (efr & 7) * 128 + ((i ^ (efr & 7)) << 4) + efg * 4) =
                            f32x2_to_bf16x2(o0, o1);
                        continue;
                    }
                    if (ACT != ACT_NONE) {
                        v0 = apply_act_c<ACT>(v0, ACT, 0.0f); v1 = apply_act_c<ACT>(v1, ACT, 0.0f);
                        v2 = apply_act_c<ACT>(v2, ACT, 0.0f); v3 = apply_act_c<ACT>(v3, ACT, 0.0f);
                    }
                    if (RESID) {
                        v0 += __uint_as_float(rv[i].x << 16);
                        v1 += __uint_as_float(rv[i].x & 0xffff0000u);
                        v2 += __uint_as_float(rv[i].y << 16);
                        v3 += __uint_as_float(rv[i].y & 0xffff0000u);
                    }
                    *reinterpret_cast<uint2 *>(slot[i]) = make_uint2(f32x2_to_bf16x2(v0, v1), f32x2_to_bf16x2(v2, v3));
                }
                if (ACT == ACT_SILU_GLU) {                 // 64 outputs per row: 128 bytes, eight 16-byte pieces; 8 rows per instruction
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const int r8 = el >> 3, ch = el & 7;
                        *reinterpret_cast<uint4 *>(p.out + (int64_t)(mrow0 + j * 16 + it * 8 + r8) * p.ldc + (ncol0 >> 1) + ch * 8) =
                            *reinterpret_cast<const uint4 *>(wbuf + it * 2048 + r8 * 128 + ((ch ^ r8) << 4));
                    }
                } else {
                    // (native vectors: as `uint4` structs these four stayed a stack slot -- scratch stores and loads, each load behind an
                    // s_waitcnt vmcnt(0) that also waited for the residual pieces in flight: 66 us of a 186 us launch)
                    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                    u32x4_t ov0, ov1, ov2, ov3;
                    ov0 = *reinterpret_cast<const u32x4_t *>(wbuf + 0 * 2048 + trow * 256 + ((tch ^ (0 * 4 + trow)) << 4));
                    ov1 = *reinterpret_cast<const u32x4_t *>(wbuf + 1 * 2048 + trow * 256 + ((tch ^ (1 * 4 + trow)) << 4));
                    ov2 = *reinterpret_cast<const u32x4_t *>(wbuf + 2 * 2048 + trow * 256 + ((tch ^ (2 * 4 + trow)) << 4));
                    ov3 = *reinterpret_cast<const u32x4_t *>(wbuf + 3 * 2048 + trow * 256 + ((tch ^ (3 * 4 + trow)) << 4));
                    if (RESID) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ov0), "+v"(ov1), "+v"(ov2), "+v"(ov3)::"memory");   // the piece is in registers: its buffer may be refilled
                    if (abl & 16) {
                        asm volatile("" ::"v"(ov0), "v"(ov1), "v"(ov2), "v"(ov3));
                    } else {
                        uint16_t *ob = p.out + (int64_t)(mrow0 + j * 16 + trow) * p.ldc + ncol0 + tch * 8;
                        *reinterpret_cast<u32x4_t *>(ob) = ov0;
                        *reinterpret_cast<u32x4_t *>(ob + (int64_t)4 * p.ldc) = ov1;
                        *reinterpret_cast<u32x4_t *>(ob + (int64_t)8 * p.ldc) = ov2;
                        *reinterpret_cast<u32x4_t *>(ob + (int64_t)12 * p.ldc) = ov3;
                    }
                }
                if (RESID && j < 2) {
#pragma unroll
                    for (int it = 0; it < 4; it++) resid_dma(j + 2, it);
                }
            }
        }
        G8_STAMP(pf_e);
        if (!has_next) break;
        t = tnext;
        tpart = pnext;
        nds = tpart < 0 ? nds_full : nds_part;
        kitem++;
        cur = nxt;
        par ^= 1;
    }
#ifdef GB_DEV_ABL
    if ((abl & 8) && p.prof && tid == 0) {
        atomicAdd((unsigned long long *)p.prof + 0, (unsigned long long)pf_k);
        atomicAdd((unsigned long long *)p.prof + 1, (unsigned long long)pf_e);
        atomicAdd((unsigned long long *)p.prof + 2, (unsigned long long)pf_b);
        atomicAdd((unsigned long long *)p.prof + 3, (unsigned long long)pf_n);
        atomicAdd((unsigned long long *)p.prof + 4, (unsigned long long)pf_v);
        atomicAdd((unsigned long long *)p.prof + 5, (unsigned long long)pf_s);
    }
#endif
#undef G8_STAGE
#undef G8_QUARTER
#undef G8_LD
#undef G8_DMA
#undef G8_STAMP
#undef G8_MID_STAMP
}

// The parts of the last round's tiles added in part order + the epilogue of k_gemm_big8 (bias, activation, residual in f32, one
// rounding): a thread = 4 consecutive columns of one row of one tile.
__global__ __launch_bounds__(256) void k_big8_split_finish(const GemmBig8Params p, const int tile0, const int act)
{
    const int tl = blockIdx.x >> 6, m = (blockIdx.x & 63) * 4 + (threadIdx.x >> 6), n = (threadIdx.x & 63) * 4;
    const int tile = tile0 + tl, mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const float *wp = p.split_ws + (int64_t)tl * p.split_s * 65536 + m * 256 + n;
    float4 v = *reinterpret_cast<const float4 *>(wp);
    for (int z = 1; z < p.split_s; z++) {
        const float4 u = *reinterpret_cast<const float4 *>(wp + (int64_t)z * 65536);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int64_t row = (int64_t)mt * G8_BM + m;
    const int col = nt * G8_BN + n;
    if (p.bias) {
        const float4 b = *reinterpret_cast<const float4 *>(p.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (act == ACT_SILU_GLU) {           // interleaved (gate, up) columns: two outputs of the half-width result (k_gemm_big8's arithmetic)
        const float o0 = v.x / (1.0f + __expf(-v.x)) * v.y, o1 = v.z / (1.0f + __expf(-v.z)) * v.w;
        *reinterpret_cast<uint32_t *>(p.out + row * p.ldc + (col >> 1)) = f32x2_to_bf16x2(o0, o1);
        return;
    }
    if (act != ACT_NONE) {
        v.x = apply_act(v.x, act, 0.0f); v.y = apply_act(v.y, act, 0.0f); v.z = apply_act(v.z, act, 0.0f); v.w = apply_act(v.w, act, 0.0f);
    }
    if (p.resid) {
        const uint2 rv = *reinterpret_cast<const uint2 *>(p.resid + row * p.ldr + col);
        v.x += __uint_as_float(rv.x << 16);
        v.y += __uint_as_float(rv.x & 0xffff0000u);
        v.z += __uint_as_float(rv.y << 16);
        v.w += __uint_as_float(rv.y & 0xffff0000u);
    }
    *reinterpret_cast<uint2 *>(p.out + row * p.ldc + col) = make_uint2(f32x2_to_bf16x2(v.x, v.y), f32x2_to_bf16x2(v.z, v.w));
}

// true if it took the launch (try_launch_gemm_big has checked the epilogue and the views; here: whole 256 x 256 tiles, K in 128s)
bool try_launch_gemm_big8(const IgemmParams &p, int64_t M, hipStream_t st, float *ws, int64_t ws_floats)
{
    constexpr int on = 1;      // fixed by measurement (profiles/NOTES.md): 0 = the 256 x 128 kernel
    if (!on || M % G8_BM || p.N % G8_BN || p.K % (2 * G8_DK) || p.K < 4 * G8_DK) return false;
    GemmBig8Params g;
    g.x = p.x; g.lda = p.lda; g.w = p.w; g.bias = p.bias; g.resid = p.resid; g.ldr = p.resid_ld;
    g.out = (uint16_t *)p.out; g.ldc = p.ldc; g.M = (int)M; g.N = p.N; g.K = p.K;
    g.mtiles = (int)(M / G8_BM); g.ntiles = p.N / G8_BN;
    g.prof = nullptr;
    g.split_ws = nullptr;
    g.split_s = 1;
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
    // one workgroup per CU of the budget the process has set for persistent kernels (ifh_set_cu_budget: inside the speech pipeline the
    // encoder's products measured 1.7 % better on the vocoder's 160 than on all 256), or of the device if the caller says it runs alone
    // (ifh_conv_desc.whole_chip: the LLM's prompt pass -- C5 share turn 129.6 -> 120 ms); IFH_GEMM_BIG8_CUS (tuning switch): that many
    constexpr int cus_env = 0;
    int grid = (cus_env > 0 ? cus_env : (p.whole_chip ? device_cu_count_physical() : device_cu_count())) & ~7;
    if (grid < 8) grid = 8;
    if (grid > total) grid = total < 8 ? total : (total & ~7);
    // with fewer than 8 tiles the XCD interleave below degenerates: one workgroup per tile
    void *args[] = {(void *)&g};
    if (total < 8) {
        // first = (b & 7) * (G >> 3) + (b >> 3) needs G >= 8; run as 8 workgroups, the surplus ones return at once
        grid = 8;
    }
    // The last round: with L = total % grid tiles left for at most half of the workgroups, and a workspace of the caller's, those tiles
    // are cut into S parts of K (S: the largest divisor of the stage count with an even part of >= 4 stages and L S <= grid) + a
    // finishing pass; 288 tiles of the LLM prompt's down projection on 256 CUs: 2 tile times -> 1.14 + the pass.
    int nleft = 0;
    if (ws && (((uintptr_t)ws) & 15) == 0 && total > grid) {
        constexpr int split_on = 1;      // fixed by measurement (profiles/NOTES.md)
        const int L = total % grid, nst = g.K / G8_DK;
        if (split_on && L > 0 && 2 * L <= grid) {
            int S = 1;
            for (int s_ = 2; s_ * L <= grid && s_ <= nst / 4; s_++)
                if (nst % s_ == 0 && (nst / s_) % 2 == 0 && (int64_t)L * s_ * 65536 <= ws_floats) S = s_;
            if (S > 2) {            // (two parts: the finishing pass costs what the half round saves)
                g.split_ws = ws;
                g.split_s = S;
                nleft = L;
            }
        }
    }
    bool ok = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(512), args, G8_LDS, st) == hipSuccess;
    if (ok && nleft)
        hipLaunchKernelGGL(k_big8_split_finish, dim3((unsigned)(nleft * 64)), dim3(256), 0, st, g, total - nleft, p.act);
#ifdef GB_DEV_ABL
    if (ok && g.prof) {
        long long h[6] = {0, 0, 0, 0, 0, 0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, g.prof, sizeof h, hipMemcpyDeviceToHost);
        if (h[3])
            fprintf(stderr, "k_gemm_big8 %d x %d x %d: per tile (wave 0, shader clocks): K loop %.1f (of it: DMA + LDS waits in front of the stage barriers %.1f, the barriers %.1f)  epilogue %.1f  entry barrier %.1f   (%lld tiles, grid %d)\n",
                    g.M, g.N, g.K, (double)h[0] / h[3], (double)h[4] / h[3], (double)h[5] / h[3], (double)h[1] / h[3], (double)h[2] / h[3], h[3], grid);
    }
#endif
    return ok;
}

}  // namespace ifh
