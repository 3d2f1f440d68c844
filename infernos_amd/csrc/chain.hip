// chain.hip -- one whole HiFi-GAN residual block in a single launch:
//     for d in (1, 3, 5):  x = x + conv_k,1(lrelu(conv_k,d(lrelu(x))))          (k = taps, "same" padding)
//     out = x * out_scale  (+ out)
// (transformers modeling_speecht5.py HifiGanResidualBlock.forward, called three times per upsampling level from
// SpeechT5HifiGan.forward with the mean over the three blocks; reached from HelloSippyTTSRT/HelloSippyRTPipe.py:236).
//
// Why: launched pair by pair (resblock.hip) every residual pair reads its input tile from HBM and writes its output back,
// a block lives ~10 us for ~0.5 us of matrix work at C <= 64 (load -> barrier -> conv -> barrier -> conv -> store, nothing
// to overlap the latencies with), and every block re-streams the weights through a register prefetch with two barriers
// per 64..128-wide chunk.  Here
//   * the activation tile stays in LDS for all six convolutions of the block: x is read once and the result written once
//     (rows recomputed on either side of a tile: 60 / 36 / 12 for k = 11 / 7 / 3);
//   * the raw residual stream never touches LDS: the lane that computes an output element of one pair is the lane that
//     adds it as the residual of the next (accumulator layout, packed bf16 in registers); LDS only holds the two
//     LeakyReLU'd operand images (x and the intermediate), each rounded to bf16 exactly where the separate launches
//     round, so the result is bit-identical to the pair-by-pair path;
//   * workgroups are persistent (one per CU, 8 waves) and fetch the next tile's rows into registers while the current
//     tile computes;
//   * the weights of the six convolutions form ONE stream of pre-packed MFMA fragments (host: ops.w_chain_pack) that is
//     DMA'd global -> LDS (global_load_lds, 16 B per lane, no VGPR round trip) into a small ring, D units ahead, with one
//     barrier per 8 KB unit; a fragment is 64 lanes x 16 B in lane order, so fragment reads are conflict-free
//     by construction and a unit is exactly one DMA instruction per wave (uniform vmcnt bookkeeping);
//   * activation rows are C*2 + 32 bytes apart: with the row stride = 2 (mod 4) sixteen-byte slots, the four lane
//     groups of a ds_read_b128 fragment read (16 rows x 4 k-groups) fall on 16 distinct slots each -- conflict-free.
//
// MFMA: v_mfma_f32_16x16x32_bf16, A = weights (16 output channels x 32 k), B = activation rows (32 k x 16 rows), k runs
// tap-major then channel, ascending -- the accumulation order of resblock.hip / conv.hip.
#include <stdlib.h>

#include <type_traits>

#include "igemm.h"
#include "chain_util.h"

// CHAIN_ABL (tools/ builds only, wrong results): 1 = no MFMAs, 2 = no fragment reads, 4 = no weight-unit sync
#ifndef CHAIN_ABL
#define CHAIN_ABL 0
#endif

namespace ifh {

struct ChainParams {
    const uint16_t *x;
    int64_t x_bstride;
    const uint16_t *wstream;     // packed fragments of the 6 convolutions, padded to whole 8 KB units
    const float *bias;           // [6][C]
    int T, nbatch;
    int tiles_per_seq, ntiles;
    int nunits;                  // units in the stream (one pass of the chain)
    float slope, out_scale;
    float post_slope;            // LeakyReLU on what is stored (1 = none): ifh_chain_desc.post_slope
    int accumulate;
    uint16_t *out;
    int64_t out_bstride;
    unsigned long long *prof;    // optional diagnostic: shader-clock sums per phase (wave 0 of every block), see ifh_chain_desc
    int exp;                     // PROF builds only (env IFH_CHAIN_EXP): ablations -- 1 no unit sync, 2 no fragment reads, 4 no MFMAs (wrong results)
};

constexpr int kUnitBytes = 8192;     // 8 waves x 64 lanes x 16 B: one global_load_lds per wave
constexpr int kGuardM = 5;           // rows an 11-tap plain convolution reaches beyond its outputs

// C channels; TAPS; waves = WGM (row groups) x WGN (channel groups) = 8; a wave owns MT row tiles x NT channel tiles of
// 16 x 16; HC = rows computed on either side of the R = WGM*MT*16 - 2*HC output rows of a tile; NRING = 8 KB weight units
// the ring holds (3 or 4); GX = guard rows of the x image (>= 5 * (TAPS-1)/2).
template <int C, int TAPS, int WGM, int WGN, int MT, int NT, int HC, int NRING, int GX, bool PROF>
__global__ __launch_bounds__(512, 2) void k_resblock_chain(const ChainParams p)
{
    static_assert(WGM * WGN == 8, "8 waves");
    static_assert(WGN * NT * 16 == C, "a block covers every channel");
    static_assert(GX >= 5 * (TAPS - 1) / 2 && (NRING == 3 || NRING == 4), "guard rows / ring depth");
    constexpr int XS = C + 16;                         // row stride in elements (C*2 + 32 bytes)
    constexpr int SB = XS * 2;                         // ... in bytes
    constexpr int RT = WGM * MT * 16;                  // rows computed per tile
    constexpr int R = RT - 2 * HC;                     // rows stored per tile
    constexpr int XROWS = RT + 2 * GX, MROWS = RT + 2 * kGuardM;
    constexpr int KSUB = C / 32;                       // k-steps per tap
    constexpr int KS = TAPS * KSUB;                    // k-steps per convolution
    constexpr int FRAGS = C / 16;                      // 1 KB fragments per k-step
    constexpr int UK = kUnitBytes / (FRAGS * 1024);    // k-steps per unit
    static_assert(UK >= 1 && UK * FRAGS * 1024 == kUnitBytes, "a unit is a whole number of k-steps");
    constexpr int XL_OFF = 0;
    constexpr int M_OFF = XL_OFF + XROWS * SB;
    constexpr int RING_OFF = M_OFF + MROWS * SB;
    constexpr int BIAS_OFF = RING_OFF + NRING * kUnitBytes;
    constexpr int H = (TAPS - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int wm = wid % WGM, wn = wid / WGM;

    // ---- once per block: zero both operand images (guard rows stay zero for good), biases to LDS, first weight units
    for (int i = tid * 16; i < RING_OFF; i += 512 * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < 6 * C; i += 512) reinterpret_cast<float *>(lds + BIAS_OFF)[i] = p.bias[i];
    const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(p.wstream) + wid * 1024 + lane * 16;
    int u_issue = 0;                  // stream unit the next DMA fetches (wraps at nunits)
    int ring_issue = 0;               // ring slot it lands in
    int ring_read = 0;                // ring slot of the unit being read
    int ks_in_unit = 0;               // k-steps of that unit already taken (0 = the next fragment read enters a new unit)
#define CHAIN_DMA()                                                                                               \
    {                                                                                                             \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + (int64_t)u_issue * kUnitBytes), \
                                         (__attribute__((address_space(3))) void *)(                              \
                                             lds + RING_OFF + ring_issue * kUnitBytes + wid * 1024), 16, 0, 0);   \
        u_issue = (u_issue + 1 == p.nunits) ? 0 : u_issue + 1;                                                    \
        ring_issue = (ring_issue + 1 == NRING) ? 0 : ring_issue + 1;                                              \
    }
#pragma unroll
    for (int d = 0; d < NRING - 1; d++) CHAIN_DMA()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // the images are zero before any wave writes its rows

    // Weight-ring protocol.  The ring holds units v .. v+NRING-1 of the stream.  ENTERING unit v (the first fragment read
    // of it) every wave (1) waits until its own piece of unit v+1 has landed (all DMAs but the NRING-3 youngest) and its
    // own LDS reads have returned, (2) meets the others at the barrier: now unit v+1 is complete for everybody -- unit v
    // already is, by the previous entry -- and nobody will read unit v-1 again (its last fragments are in registers),
    // (3) refills the slot of unit v-1 with unit v+NRING-1.  Fragment reads of unit v+1 therefore need no barrier of their
    // own and are issued ahead of the MFMAs that precede them (register double buffer below).
#define CHAIN_ENTER_UNIT()                                                         \
    {                                                                              \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRING - 3) : "memory");          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         \
        __builtin_amdgcn_s_barrier();                                              \
        CHAIN_DMA()                                                                \
    }

    // per-lane bases (bytes)
    const int row0 = wm * MT * 16 + fr;                                  // first owned row (tile-relative)
    const int xb = XL_OFF + (GX + row0) * SB + fg * 16;                  // B-fragment base in the x image
    const int mb = M_OFF + (kGuardM + row0) * SB + fg * 16;              // ... in the intermediate image
    const int ab = RING_OFF + wn * NT * 1024 + lane * 16;                // A-fragment base inside a k-step of the ring
    const int cw = (wn * NT * 16 + 4 * fg) * 2;                          // byte offset of this lane's 4 channels in a row (tile 0)
    const int bias_b = BIAS_OFF + (wn * NT * 16 + 4 * fg) * 4;           // ... of its 4 biases (convolution 0, tile 0)

    // residual stream of the owned elements, packed bf16: xr[j][i] = rows of tile j, channels of tile i
    uint2 xr[MT][NT], xn[MT][NT], pv[MT][NT];
    auto load_rows = [&](const uint16_t *base, int64_t bstride, int tile, uint2 (&dst)[MT][NT]) {
        const int b = tile / p.tiles_per_seq, ti = tile - b * p.tiles_per_seq;
        const int tq = ti * R - HC;
#pragma unroll
        for (int j = 0; j < MT; j++) {
            const int t = min(max(tq + row0 + j * 16, 0), p.T - 1);
#pragma unroll
            for (int i = 0; i < NT; i++)
                dst[j][i] = *reinterpret_cast<const uint2 *>(base + (int64_t)b * bstride + (int64_t)t * C + (wn * NT + i) * 16 + 4 * fg);
        }
    };
    int tile = blockIdx.x;
    if (tile < p.ntiles) load_rows(p.x, p.x_bstride, tile, xn);
    // diagnostic phase clocks (PROF builds, selected when ifh_chain_desc.debug_prof is set): wave 0 adds the shader clocks
    // spent in [0] tile top (residual copy, loads issued, x image written, barrier), [1 + 2q] the K loop of convolution q,
    // [2 + 2q] its epilogue + barrier; [13] tiles, [14] workgroup lifetime, [15] workgroups
    unsigned long long tprev = PROF ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long tstart = tprev;
#define CHAIN_STAMP(IDX)                                                   \
    if (PROF) {                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        if (tid == 0) atomicAdd(p.prof + (IDX), now_ - tprev);             \
        tprev = now_;                                                      \
    }

    f32x4 acc[NT][MT];
    static_assert(NT == 2, "the read/MFMA interleave below is written for two channel tiles per wave");
    constexpr int NR = NT + MT;                        // fragment reads per k-step
    // Fragments of one k-step: A = this wave's NT weight tiles (ring), B = its MT row tiles of the operand image, read in
    // the order the MFMAs want them: q0 = A0, q1..qMT = B0..B(MT-1), q(MT+1) = A1.  Every LDS access of the main loop goes
    // through inline asm with hand-counted s_waitcnt: as ordinary loads the compiler waits lgkmcnt(0) in front of the MFMAs
    // (the reads just issued for the NEXT k-step included) and vmcnt(0) in front of any LDS access while a weight DMA is
    // pending -- either one serialises the pipeline.
    int nxt_a = 0, nxt_b = 0;                          // byte addresses of the fragments to fetch next
    bool nxt_enter = false;                            // ... and whether that k-step is the first of a weight unit
    // addresses of k-step s of the current convolution; no synchronisation here, so that it can sit in the shadow of MFMAs
    auto advance = [&](int src, int dd, int s) {
        nxt_enter = ks_in_unit == 0;
        const int tap = s / KSUB, cs = s - tap * KSUB;
        nxt_a = ab + ring_read * kUnitBytes + ks_in_unit * (FRAGS * 1024);
        nxt_b = src + (tap - H) * dd * SB + cs * 64;
        if (++ks_in_unit == UK) {
            ks_in_unit = 0;
            ring_read = (ring_read + 1 == NRING) ? 0 : ring_read + 1;
        }
    };
#define CHAIN_READ(Q, FA, FB)                                                                                             \
    {                                                                                                                     \
        if ((Q) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(FA[0]) : "v"(nxt_a));                                     \
        else if ((Q) == MT + 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(FA[1]) : "v"(nxt_a));               \
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FB[(Q) - 1]) : "v"(nxt_b), "n"(((Q) - 1) * 16 * SB));    \
    }
    // One k-step: the MFMAs on the fragments in (ca, cb) in the order (A0 x B0..B(MT-1)), (A1 x B0..B(MT-1)), with the NR
    // reads of the NEXT k-step (addresses already in nxt_*) into (na, nb) issued one per MFMA in front of the first NR of
    // them -- all eight waves run in step, so reads issued as one burst would occupy the LDS for ~250 cycles in which no MFMA
    // issues.  LDS reads return in order: in front of MFMA k (k <= MT) the fragment it needs is back once at most NR - 2
    // reads are outstanding.  If the next k-step opens a weight unit the waves meet first (CHAIN_ENTER_UNIT without its
    // DMA); everything that is not needed for the first MFMA -- the refill DMA, the addresses of the k-step after next
    // (s_after, or < 0) -- is issued behind MFMAs, because between the barrier and the first MFMA the matrix pipe idles.
    auto kstep = [&](const bf16x8_t (&ca)[NT], const bf16x8_t (&cb)[MT], bf16x8_t (&na)[NT], bf16x8_t (&nb)[MT], auto has_next,
                     int src, int dd, int s_after) {
        constexpr bool NEXT = decltype(has_next)::value;
        const bool enter = NEXT && nxt_enter;
        if (enter && !(CHAIN_ABL & 4) && !(PROF && (p.exp & 1))) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRING - 3) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (!NEXT) {
            asm volatile("s_waitcnt lgkmcnt(0)");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < NT * MT; k++) {
            if (NEXT && !(CHAIN_ABL & 2) && !(PROF && (p.exp & 2))) {
                // MFMA k <= MT needs fragment q(k+1) (q0 too for k = 0) of the current set: reads return in order, so it is
                // back once no more than [reads of this set behind it] + [reads of the next set issued so far] are outstanding
                if (k <= MT) wait_lgkm(NR - 2 - (k < MT ? k : MT) + rd_before(k, NR, NT * MT) + (k == MT ? NT - 2 : 0));
                if (rd_at(k, NR, NT * MT) >= 0) CHAIN_READ(rd_at(k, NR, NT * MT), na, nb)
                __builtin_amdgcn_sched_barrier(0);
            }
            const int i = k / MT, j = k - i * MT;
            if (!(CHAIN_ABL & 1) && !(PROF && (p.exp & 4))) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k == 1 && enter) {
                CHAIN_DMA()
                __builtin_amdgcn_sched_barrier(0);
            }
            if (k == NT * MT - 1 && s_after >= 0) {      // every read of the next set has its address: move on to the one after
                advance(src, dd, s_after);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // one convolution over the image at `src` with dilation dd, two fragment sets
    auto conv = [&](int src, int dd) {
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8_t fa0[NT], fb0[MT], fa1[NT], fb1[MT];
        advance(src, dd, 0);
        if (nxt_enter) CHAIN_ENTER_UNIT()
#pragma unroll
        for (int q = 0; q < NR; q++) CHAIN_READ(q, fa0, fb0)
        __builtin_amdgcn_sched_barrier(0);
        if (KS > 1) advance(src, dd, 1);
        int s = 0;
#pragma unroll 1
        for (; s + 2 < KS; s += 2) {
            kstep(fa0, fb0, fa1, fb1, std::true_type{}, src, dd, s + 2);
            kstep(fa1, fb1, fa0, fb0, std::true_type{}, src, dd, s + 3 < KS ? s + 3 : -1);
        }
        if (s + 1 < KS) {
            kstep(fa0, fb0, fa1, fb1, std::true_type{}, src, dd, -1);
            kstep(fa1, fb1, fa0, fb0, std::false_type{}, src, dd, -1);
        } else {
            kstep(fa0, fb0, fa1, fb1, std::false_type{}, src, dd, -1);
        }
    };
    // this lane's biases of convolution q: read through inline asm -- as an ordinary LDS read the compiler puts an
    // s_waitcnt vmcnt(0) in front of it (pending LDS-DMA), which would drain the weight prefetch at every epilogue
    // epilogue stores to the operand images, through inline asm for the same reason (offset: compile-time, < 64 KB)
#define CHAIN_LDS_STORE(ADDR, OFF, VAL)                                                                              \
    {                                                                                                                \
        const uint2 v_ = (VAL);                                                                                      \
        const unsigned long long q_ = ((unsigned long long)v_.y << 32) | v_.x;                                       \
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ADDR), "v"(q_), "n"(OFF) : "memory");                     \
    }
    const int xw = XL_OFF + (GX + row0) * SB + cw, mw = M_OFF + (kGuardM + row0) * SB + cw;     // store bases of this lane
    auto read_bias = [&](int q, f32x4 (&bv)[NT]) {
        const int addr = bias_b + q * C * 4;
#pragma unroll
        for (int i = 0; i < NT; i++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[i]) : "v"(addr), "n"(i * 64));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    for (; tile < p.ntiles; tile += gridDim.x) {
        const int b = tile / p.tiles_per_seq, ti = tile - b * p.tiles_per_seq;
        const int tq = ti * R - HC;                                      // time of tile row 0
        const bool inside = tq >= 0 && tq + RT <= p.T;                   // no row of this tile is zero padding
#pragma unroll
        for (int j = 0; j < MT; j++)
#pragma unroll
            for (int i = 0; i < NT; i++) xr[j][i] = xn[j][i];
        // x image = LeakyReLU(x), zero outside the sequence (the convolutions' zero padding)
#pragma unroll
        for (int j = 0; j < MT; j++) {
            const int t = tq + row0 + j * 16;
            const bool ok = inside || (t >= 0 && t < p.T);
#pragma unroll
            for (int i = 0; i < NT; i++)
                *reinterpret_cast<uint2 *>(lds + XL_OFF + (GX + row0 + j * 16) * SB + cw + i * 32) =
                    ok ? chain_lrelu4(xr[j][i], p.slope) : make_uint2(0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        CHAIN_STAMP(0)

        // conv1 epilogue: + bias, round to bf16 (what a separate launch stores), LeakyReLU (what the next one applies on
        // load) -> intermediate image
        auto epi1 = [&](int q) {
            f32x4 bv[NT];
            read_bias(q, bv);
#pragma unroll
            for (int j = 0; j < MT; j++) {
                const int t = tq + row0 + j * 16;
                const bool ok = inside || (t >= 0 && t < p.T);
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    const f32x4 a = acc[i][j];
                    uint2 pk = make_uint2(f32x2_to_bf16x2(a[0] + bv[i][0], a[1] + bv[i][1]), f32x2_to_bf16x2(a[2] + bv[i][2], a[3] + bv[i][3]));
                    pk = chain_lrelu4(pk, p.slope);
                    CHAIN_LDS_STORE(mw, j * 16 * SB + i * 32, ok ? pk : make_uint2(0, 0))
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        // conv2 epilogue: + bias + residual -> new residual (registers) and its LeakyReLU'd image; the last pair scales,
        // accumulates and stores instead
        auto epi2 = [&](int q, auto last_c) {
            constexpr bool LAST = decltype(last_c)::value;
            f32x4 bv[NT];
            read_bias(q, bv);
#pragma unroll
            for (int j = 0; j < MT; j++) {
                const int qrow = row0 + j * 16, t = tq + qrow;
                const bool ok = inside || (t >= 0 && t < p.T);
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    const f32x4 a = acc[i][j];
                    const uint2 rv = xr[j][i];
                    float v0 = a[0] + bv[i][0], v1 = a[1] + bv[i][1], v2 = a[2] + bv[i][2], v3 = a[3] + bv[i][3];
                    v0 += __uint_as_float(rv.x << 16);
                    v1 += __uint_as_float(rv.x & 0xffff0000u);
                    v2 += __uint_as_float(rv.y << 16);
                    v3 += __uint_as_float(rv.y & 0xffff0000u);
                    if (LAST) {
                        v0 *= p.out_scale; v1 *= p.out_scale; v2 *= p.out_scale; v3 *= p.out_scale;
                        if (p.accumulate) {
                            const uint2 q2 = pv[j][i];
                            v0 += __uint_as_float(q2.x << 16);
                            v1 += __uint_as_float(q2.x & 0xffff0000u);
                            v2 += __uint_as_float(q2.y << 16);
                            v3 += __uint_as_float(q2.y & 0xffff0000u);
                        }
                    }
                    uint2 pk = make_uint2(f32x2_to_bf16x2(v0, v1), f32x2_to_bf16x2(v2, v3));
                    if (!LAST) {
                        xr[j][i] = pk;
                        CHAIN_LDS_STORE(xw, j * 16 * SB + i * 32, ok ? chain_lrelu4(pk, p.slope) : make_uint2(0, 0))
                    } else if (ok && qrow >= HC && qrow < HC + R) {
                        if (p.post_slope != 1.0f) pk = chain_lrelu4(pk, p.post_slope);    // (the consumer's LeakyReLU-on-load, taken here: same bits)
                        *reinterpret_cast<uint2 *>(p.out + (int64_t)b * p.out_bstride + (int64_t)t * C + (wn * NT + i) * 16 + 4 * fg) = pk;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };

#pragma unroll 1
        for (int pr = 0; pr < 2; pr++) {
            conv(xb, 2 * pr + 1);
            CHAIN_STAMP(1 + 4 * pr)
            epi1(2 * pr);
            CHAIN_STAMP(2 + 4 * pr)
            conv(mb, 1);
            CHAIN_STAMP(3 + 4 * pr)
            epi2(2 * pr + 1, std::false_type{});
            CHAIN_STAMP(4 + 4 * pr)
        }
        conv(xb, 5);
        CHAIN_STAMP(9)
        // operands of the tile end / of the next tile.  Vector memory operations retire in order, so the weight DMAs issued
        // behind these loads are not seen to land before they do: issue them where no new weight unit is needed for a
        // while -- ahead of an epilogue -- and they are back by the time the last convolution enters its first unit
        if (p.accumulate) load_rows(p.out, p.out_bstride, tile, pv);
        if (tile + (int)gridDim.x < p.ntiles) load_rows(p.x, p.x_bstride, tile + gridDim.x, xn);
        epi1(4);
        CHAIN_STAMP(10)
        conv(mb, 1);
        CHAIN_STAMP(11)
        epi2(5, std::true_type{});
        CHAIN_STAMP(12)
        if (PROF && tid == 0) atomicAdd(p.prof + 13, 1ull);
        // the weight stream restarts with every tile: what is left of a partly read unit is padding
        if (ks_in_unit != 0) {
            ks_in_unit = 0;
            ring_read = (ring_read + 1 == NRING) ? 0 : ring_read + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PROF && tid == 0) {
        atomicAdd(p.prof + 14, __builtin_amdgcn_s_memtime() - tstart);
        atomicAdd(p.prof + 15, 1ull);
    }
#undef CHAIN_STAMP
#undef CHAIN_READ
#undef CHAIN_LDS_STORE
#undef CHAIN_ENTER_UNIT
#undef CHAIN_DMA
}

template <int C, int TAPS, int WGM, int WGN, int MT, int NT, int HC, int NRING, int GX>
static int launch_chain(ChainParams &p, hipStream_t st)
{
    constexpr int XS = C + 16, SB = XS * 2, RT = WGM * MT * 16, R = RT - 2 * HC;
    constexpr int FRAGS = C / 16, UK = kUnitBytes / (FRAGS * 1024);
    static_assert(HC >= 12 * (TAPS - 1) / 2 || HC == 0, "margin covers the chain: (1+3+5 dilated + 3 plain) * (taps-1)/2 rows");
    constexpr size_t bytes = (size_t)(RT + 2 * GX) * SB + (size_t)(RT + 2 * kGuardM) * SB + (size_t)NRING * kUnitBytes + 6 * C * sizeof(float);
    static_assert(bytes <= 160 * 1024, "tile does not fit in LDS");
    if (HC == 0 && p.T > R) return fail(IFH_EINVAL, "resblock_chain: whole-sequence tile, t too large");
    const int ksteps = 6 * TAPS * (C / 32);
    if (p.nunits != (ksteps + UK - 1) / UK) return fail(IFH_EINVAL, "resblock_chain: weight stream length does not match c/taps");
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_resblock_chain<C, TAPS, WGM, WGN, MT, NT, HC, NRING, GX, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_resblock_chain<C, TAPS, WGM, WGN, MT, NT, HC, NRING, GX, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return check_hip(e, "resblock_chain lds attr");
        attr_once.done(attr_dev);
    }
    p.tiles_per_seq = (p.T + R - 1) / R;
    p.ntiles = p.tiles_per_seq * p.nbatch;
    const int ncu = device_cu_count();
    if (ncu <= 0) return fail(IFH_EHIP, "resblock_chain: device query");
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    if (p.prof)
        hipLaunchKernelGGL((k_resblock_chain<C, TAPS, WGM, WGN, MT, NT, HC, NRING, GX, true>), dim3(grid), dim3(512), bytes, st, p);
    else
        hipLaunchKernelGGL((k_resblock_chain<C, TAPS, WGM, WGN, MT, NT, HC, NRING, GX, false>), dim3(grid), dim3(512), bytes, st, p);
    return IFH_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// k_conv_ring256 -- one stride-1 convolution with C = Cin = Cout = 256 on short sequences (the first HiFi-GAN level:
// T = 48 rows per chunk), out = ((conv(lrelu(x); taps, dil) + bias) [+ resid]) * scale [+ out].
//
// At C = 256 a residual pair does not fit the chain kernel (two operand images of even ONE extra chunk exceed the LDS), and
// the round-1 kernel (conv.hip, 48..96 rows per workgroup, weights through a register prefetch) is bound by the L2 -> CU
// weight stream: every workgroup re-reads all of W (0.4-1.4 MB) for 48 rows.  Here a workgroup takes TWO chunks -- the x
// image holds [25 zero rows | chunk a | 25 zero rows | chunk b | zero rows], so the gap is both chunks' zero padding -- as
// 128 rows x 256 channels of MFMA work (8 waves x 32 output channels x 8 row tiles; 75 % of the rows are real), and the
// weights arrive through the same DMA ring as above: a k-step is 16 fragments = one 16 KB unit = two global_load_lds per
// wave, four units in the ring, fragment reads of unit v+1 never wait.  Workgroups are persistent over the chunk pairs.
struct Ring256Params {
    const uint16_t *x;
    int64_t x_bstride;
    const uint16_t *wstream;     // packed fragments of the convolution: taps * 8 units of 16 KB
    const float *bias;           // [256] or null
    const uint16_t *resid;       // [nbatch][T][256] or null
    int64_t resid_bstride;
    int T, nbatch, taps, dil;
    float pre_slope, out_scale;
    int accumulate;
    uint16_t *out;
    int64_t out_bstride;
};

// NCH chunks per workgroup: 2 (MT = 8 row tiles, 25-row gaps: any convolution of the level) or -- round 3 -- 3 (MT = 10, 8-row
// gaps) for the 14 of the level's 18 convolutions whose reach (taps - 1) / 2 * dil is at most 8: 144 of 160 MFMA rows are real
// instead of 96 of 128, and every weight unit streamed from L2 serves three chunks instead of two.
template <int MT, int NCH, int GX>
__global__ __launch_bounds__(512, 2) void k_conv_ring256(const Ring256Params p)
{
    // 8 waves = 2 row groups x 4 channel groups: a wave owns MTW = MT / 2 row tiles x NT = 4 channel tiles (64 channels).  (Rounds 2 and
    // early 3: 1 x 8, every wave all MT row tiles x 2 channel tiles = 10-12 fragment reads per 16-20 MFMAs, 160 B of LDS reads per
    // clock at full MFMA rate; now 8-9 reads, 128-144 B/clock.  Same k order per output element: same bits.)
    constexpr int C = 256, NT = 4, MTW = MT / 2, NR = NT + MTW, NRING = 4;
    static_assert(MT % 2 == 0, "two row groups");
    constexpr int SB = (C + 16) * 2;                   // 544-byte rows: stride = 2 (mod 4) sixteen-byte slots
    constexpr int RT = MT * 16;                        // 128 rows computed
    constexpr int XROWS = RT + 2 * GX;
    constexpr int UNIT = 16384;                        // one k-step: 16 fragments of 1 KB
    constexpr int RING_OFF = XROWS * SB;
    constexpr int BIAS_OFF = RING_OFF + NRING * UNIT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid & 1, wn = wid >> 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int T = p.T, pitch = T + GX;                 // chunk b starts `pitch` rows after chunk a
    const int H = (p.taps - 1) / 2, KS = p.taps * 8;

    for (int i = tid * 16; i < RING_OFF; i += 512 * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < C; i += 512) reinterpret_cast<float *>(lds + BIAS_OFF)[i] = p.bias ? p.bias[i] : 0.0f;
    const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(p.wstream) + wid * 2048 + lane * 16;
    int u_issue = 0, ring_issue = 0, ring_read = 0;
#define RING_DMA()                                                                                                       \
    {                                                                                                                    \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; h_++)                                                                 \
            __builtin_amdgcn_global_load_lds(                                                                            \
                (const __attribute__((address_space(1))) void *)(wsrc + (int64_t)u_issue * UNIT + h_ * 1024),            \
                (__attribute__((address_space(3))) void *)(lds + RING_OFF + ring_issue * UNIT + wid * 2048 + h_ * 1024), 16, 0, 0); \
        u_issue = (u_issue + 1 == KS) ? 0 : u_issue + 1;                                                                 \
        ring_issue = (ring_issue + 1 == NRING) ? 0 : ring_issue + 1;                                                     \
    }
#pragma unroll
    for (int d = 0; d < NRING - 1; d++) RING_DMA()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int ab = RING_OFF + wn * NT * 1024 + lane * 16;                // channel group wn owns output channels 64 wn .. 64 wn + 63
    const int xb = (GX + wm * MTW * 16 + fr) * SB + fg * 16;             // row group wm owns row tiles wm MTW .. wm MTW + MTW - 1
    const int bias_b = BIAS_OFF + (wn * NT * 16 + 4 * fg) * 4;
    f32x4 acc[NT][MTW];
    int nxt_a = 0, nxt_b = 0;
    // addresses of k-step s (= weight unit s): no synchronisation, so it can sit behind MFMAs
    auto advance = [&](int s) {
        const int tap = s >> 3, cs = s & 7;
        nxt_a = ab + ring_read * UNIT;
        nxt_b = xb + (tap - H) * p.dil * SB + cs * 64;
        ring_read = (ring_read + 1 == NRING) ? 0 : ring_read + 1;
    };
#define RING_READ(Q, FA, FB)                                                                                              \
    {                                                                                                                     \
        /* read order = the order the MFMAs want them: A0, B0 .. B(MTW-1), A1, A2, A3 */                                  \
        if ((Q) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(FA[0]) : "v"(nxt_a));                                     \
        else if ((Q) <= MTW) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FB[(Q) - 1]) : "v"(nxt_b), "n"(((Q) - 1) * 16 * SB)); \
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FA[(Q) - MTW]) : "v"(nxt_a), "n"(((Q) - MTW) * 1024));    \
    }
    // entering a unit (= a k-step here): this wave's two pieces of the NEXT unit have landed (all DMAs but the youngest
    // unit's two), everybody's have after the barrier; the slot of the previous unit is refilled three units ahead, behind
    // the first MFMAs (as in k_resblock_chain: nothing but the barrier sits between two k-steps' MFMAs)
#define RING_ENTER()                                            \
    {                                                           \
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        __builtin_amdgcn_s_barrier();                           \
    }
    auto kstep = [&](const bf16x8_t (&ca)[NT], const bf16x8_t (&cb)[MTW], bf16x8_t (&na)[NT], bf16x8_t (&nb)[MTW], auto has_next,
                     int s_after) {
        constexpr bool NEXT = decltype(has_next)::value;
        if (NEXT) RING_ENTER()
        if (!NEXT) {
            asm volatile("s_waitcnt lgkmcnt(0)");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < NT * MTW; k++) {
            if (NEXT) {
                // MFMA k needs fragment k + 1 (B_k) while it walks the first channel tile, fragment MTW + i (A_i) when it enters
                // channel tile i; the reads were issued in that order, so NR - 1 - (that index) of this k-step's may still be out,
                // plus those of the next k-step issued so far
                if (k < MTW) wait_lgkm(NR - 2 - k + rd_before(k, NR, NT * MTW));
                else if (k % MTW == 0) wait_lgkm(NR - 1 - (MTW + k / MTW) + rd_before(k, NR, NT * MTW));
                if (rd_at(k, NR, NT * MTW) >= 0) RING_READ(rd_at(k, NR, NT * MTW), na, nb)
                __builtin_amdgcn_sched_barrier(0);
            }
            const int i = k / MTW, j = k - i * MTW;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k == 1 && NEXT) {
                RING_DMA()
                __builtin_amdgcn_sched_barrier(0);
            }
            if (k == NT * MTW - 1 && s_after >= 0) {
                advance(s_after);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    const int ngroups = (p.nbatch + NCH - 1) / NCH;
    for (int tile = blockIdx.x; tile < ngroups; tile += gridDim.x) {
        const int b0 = tile * NCH;
        const int nb = min(NCH, p.nbatch - b0);
        // x image: rows of chunk e at GX + e * pitch + t, LeakyReLU applied once here, 16 bytes per thread per step.  (Issuing all of
        // a thread's loads ahead of the first LDS write, and the epilogue's residual loads four row tiles at a time, was measured in
        // round 3: nothing at two chunks, 8-15 % slower at three, where the extra live registers went to scratch.)
        for (int v = tid; v < nb * T * (C / 8); v += 512) {
            const int rowi = v / (C / 8), c8 = v - rowi * (C / 8);
            const int e = rowi / T, t = rowi - e * T;
            const uint4 xv = *reinterpret_cast<const uint4 *>(p.x + (int64_t)(b0 + e) * p.x_bstride + (int64_t)t * C + c8 * 8);
            *reinterpret_cast<uint4 *>(lds + (GX + e * pitch + t) * SB + c8 * 16) = p.pre_slope != 1.0f ? lrelu8(xv, p.pre_slope) : xv;
        }
        if (nb < NCH)                                  // a short last group: the missing chunks' rows of the previous tile are stale
            for (int v = tid; v < (NCH - nb) * T * (C / 8); v += 512) {
                const int rowi = v / (C / 8);
                const int e = nb + rowi / T, t = rowi - (rowi / T) * T;
                *reinterpret_cast<uint4 *>(lds + (GX + e * pitch + t) * SB + (v % (C / 8)) * 16) = make_uint4(0, 0, 0, 0);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int j = 0; j < MTW; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8_t fa0[NT], fb0[MTW], fa1[NT], fb1[MTW];
        advance(0);
        RING_ENTER()
        RING_DMA()
#pragma unroll
        for (int q = 0; q < NR; q++) RING_READ(q, fa0, fb0)
        __builtin_amdgcn_sched_barrier(0);
        advance(1);
        int s = 0;
#pragma unroll 1
        for (; s + 2 < KS; s += 2) {                   // KS = 8 * taps is even
            kstep(fa0, fb0, fa1, fb1, std::true_type{}, s + 2);
            kstep(fa1, fb1, fa0, fb0, std::true_type{}, s + 3);
        }
        kstep(fa0, fb0, fa1, fb1, std::true_type{}, -1);
        kstep(fa1, fb1, fa0, fb0, std::false_type{}, -1);
        // epilogue: + bias (+ residual) * scale (+ previous out) -> global, 8 bytes per lane
        f32x4 bv[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[i]) : "v"(bias_b), "n"(i * 64));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < MTW; j++) {
            const int q = (wm * MTW + j) * 16 + fr;
            const int e = q / pitch, t = q - e * pitch;
            if (t < T && e < nb) {
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    const int64_t off = (int64_t)t * C + (wn * NT + i) * 16 + 4 * fg;
                    const f32x4 a = acc[i][j];
                    float v0 = a[0] + bv[i][0], v1 = a[1] + bv[i][1], v2 = a[2] + bv[i][2], v3 = a[3] + bv[i][3];
                    if (p.resid) {
                        const uint2 rv = *reinterpret_cast<const uint2 *>(p.resid + (int64_t)(b0 + e) * p.resid_bstride + off);
                        v0 += __uint_as_float(rv.x << 16);
                        v1 += __uint_as_float(rv.x & 0xffff0000u);
                        v2 += __uint_as_float(rv.y << 16);
                        v3 += __uint_as_float(rv.y & 0xffff0000u);
                    }
                    v0 *= p.out_scale; v1 *= p.out_scale; v2 *= p.out_scale; v3 *= p.out_scale;
                    uint16_t *dst = p.out + (int64_t)(b0 + e) * p.out_bstride + off;
                    if (p.accumulate) {
                        const uint2 q2 = *reinterpret_cast<const uint2 *>(dst);
                        v0 += __uint_as_float(q2.x << 16);
                        v1 += __uint_as_float(q2.x & 0xffff0000u);
                        v2 += __uint_as_float(q2.y << 16);
                        v3 += __uint_as_float(q2.y & 0xffff0000u);
                    }
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(f32x2_to_bf16x2(v0, v1), f32x2_to_bf16x2(v2, v3));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave is done with the x image before the next tile overwrites it
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef RING_ENTER
#undef RING_READ
#undef RING_DMA
}

}  // namespace ifh

using namespace ifh;

extern "C" int ifh_resblock_chain_bf16(const ifh_chain_desc *d, ifh_stream_t stream)
{
    IFH_CHECK_ARG(d);
    IFH_CHECK_ARG(d->x && d->wstream && d->bias && d->out);
    IFH_CHECK_ARG(d->nbatch >= 0 && d->t >= 0);
    if (d->nbatch == 0 || d->t == 0) return IFH_OK;
    IFH_CHECK_ARG(d->c == 32 || d->c == 64 || d->c == 128);
    IFH_CHECK_ARG(d->taps == 3 || d->taps == 7 || d->taps == 11);
    IFH_CHECK_ARG((((uintptr_t)d->x) & 7) == 0 && (((uintptr_t)d->out) & 7) == 0 && (((uintptr_t)d->wstream) & 15) == 0 &&
                  (((uintptr_t)d->bias) & 3) == 0 && d->x_bstride % 4 == 0 && d->out_bstride % 4 == 0);
    IFH_CHECK_ARG(d->slope > 0.0f && d->slope <= 1.0f && (int64_t)d->nbatch * d->t < (1ll << 31));
    ChainParams p;
    p.x = (const uint16_t *)d->x;
    p.x_bstride = d->x_bstride;
    p.wstream = (const uint16_t *)d->wstream;
    p.bias = d->bias;
    p.T = d->t;
    p.nbatch = d->nbatch;
    p.nunits = d->nunits;
    p.slope = d->slope;
    p.out_scale = d->out_scale;
    IFH_CHECK_ARG(d->post_slope >= 0.0f && d->post_slope <= 1.0f);
    p.post_slope = d->post_slope > 0.0f ? d->post_slope : 1.0f;      // (0 = a zeroed descriptor: none)
    p.accumulate = d->accumulate;
    p.out = (uint16_t *)d->out;
    p.out_bstride = d->out_bstride;
    p.prof = (unsigned long long *)d->debug_prof;
    p.exp = (p.prof && getenv("IFH_CHAIN_EXP")) ? atoi(getenv("IFH_CHAIN_EXP")) : 0;
    hipStream_t st = as_stream(stream);
    int rc = IFH_EINVAL;
    //                      <C, TAPS, WGM, WGN, MT, NT, HC, NRING, GX>
#define CHAIN_CASE(C_, K_, ...)                      \
    if (d->c == C_ && d->taps == K_) rc = launch_chain<C_, K_, __VA_ARGS__>(p, st);
    // C = 32: 640 rows computed, 512 (k = 11), 544 (k = 7), 608 (k = 3) stored per tile
    CHAIN_CASE(32, 11, 8, 1, 5, 2, 64, 4, 25)
    CHAIN_CASE(32, 7, 8, 1, 5, 2, 48, 4, 15)
    CHAIN_CASE(32, 3, 8, 1, 5, 2, 16, 4, 5)
    // C = 64: 384 rows computed and 256 / 288 stored (k = 11 / 7); 320 computed, 256 stored (k = 3)
    CHAIN_CASE(64, 11, 4, 2, 6, 2, 64, 3, 25)
    CHAIN_CASE(64, 7, 4, 2, 6, 2, 48, 4, 15)
    CHAIN_CASE(64, 3, 4, 2, 5, 2, 32, 4, 5)
    // C = 128: the whole 192-row sequence of a chunk is one tile (nothing recomputed)
    if (d->c == 128) {
        IFH_CHECK_ARG(d->t <= 192);
        CHAIN_CASE(128, 11, 2, 4, 6, 2, 0, 4, 25)
        CHAIN_CASE(128, 7, 2, 4, 6, 2, 0, 4, 15)
        CHAIN_CASE(128, 3, 2, 4, 6, 2, 0, 4, 5)
    }
#undef CHAIN_CASE
    if (rc != IFH_OK) return rc;
    IFH_LAUNCH_CHECK("resblock_chain_bf16");
    return IFH_OK;
}

extern "C" int ifh_conv_ring256_bf16(const ifh_ring256_desc *d, ifh_stream_t stream)
{
    IFH_CHECK_ARG(d);
    IFH_CHECK_ARG(d->x && d->wstream && d->out);
    IFH_CHECK_ARG(d->nbatch >= 0 && d->t >= 0);
    if (d->nbatch == 0 || d->t == 0) return IFH_OK;
    IFH_CHECK_ARG(d->t <= 48 && d->taps >= 1 && d->taps <= 11 && (d->taps & 1) == 1 && d->dil >= 1 && (d->taps - 1) / 2 * d->dil <= 25);
    IFH_CHECK_ARG((((uintptr_t)d->x) & 15) == 0 && (((uintptr_t)d->out) & 7) == 0 && (((uintptr_t)d->wstream) & 15) == 0 &&
                  (!d->resid || (((uintptr_t)d->resid) & 7) == 0) && (!d->bias || (((uintptr_t)d->bias) & 3) == 0) &&
                  d->x_bstride % 8 == 0 && d->out_bstride % 4 == 0 && d->resid_bstride % 4 == 0);
    IFH_CHECK_ARG(d->pre_slope > 0.0f && d->pre_slope <= 1.0f);
    Ring256Params p;
    p.x = (const uint16_t *)d->x;
    p.x_bstride = d->x_bstride;
    p.wstream = (const uint16_t *)d->wstream;
    p.bias = d->bias;
    p.resid = (const uint16_t *)d->resid;
    p.resid_bstride = d->resid_bstride;
    p.T = d->t;
    p.nbatch = d->nbatch;
    p.taps = d->taps;
    p.dil = d->dil;
    p.pre_slope = d->pre_slope;
    p.out_scale = d->out_scale;
    p.accumulate = d->accumulate;
    p.out = (uint16_t *)d->out;
    p.out_bstride = d->out_bstride;
    const int ncu = device_cu_count();
    if (ncu <= 0) return fail(IFH_EHIP, "conv_ring256: device query");
    const int reach = (d->taps - 1) / 2 * d->dil;
    constexpr bool no3 = false;            // fixed by measurement (profiles/NOTES.md)
    if (reach <= 8 && !no3 && d->nbatch >= 3 * ncu) {
        constexpr size_t bytes = (size_t)(160 + 16) * 544 + 4 * 16384 + 256 * sizeof(float);
        static_assert(bytes <= 160 * 1024, "ring256 tile");
        static_assert(3 * 48 + 2 * 8 <= 160, "three chunks and their gaps fit the row tiles");
        static DeviceOnce attr_once;
        int attr_dev = 0;
        if (attr_once.needed(&attr_dev)) {
            hipError_t e = hipFuncSetAttribute((const void *)k_conv_ring256<10, 3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return check_hip(e, "conv_ring256 lds attr");
            attr_once.done(attr_dev);
        }
        const int ng = (d->nbatch + 2) / 3;
        hipLaunchKernelGGL((k_conv_ring256<10, 3, 8>), dim3(ng < ncu ? ng : ncu), dim3(512), bytes, as_stream(stream), p);
    } else {
        constexpr size_t bytes = (size_t)(128 + 50) * 544 + 4 * 16384 + 256 * sizeof(float);
        static_assert(bytes <= 160 * 1024, "ring256 tile");
        static DeviceOnce attr_once;
        int attr_dev = 0;
        if (attr_once.needed(&attr_dev)) {
            hipError_t e = hipFuncSetAttribute((const void *)k_conv_ring256<8, 2, 25>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return check_hip(e, "conv_ring256 lds attr");
            attr_once.done(attr_dev);
        }
        const int npairs = (d->nbatch + 1) / 2;
        hipLaunchKernelGGL((k_conv_ring256<8, 2, 25>), dim3(npairs < ncu ? npairs : ncu), dim3(512), bytes, as_stream(stream), p);
    }
    IFH_LAUNCH_CHECK("conv_ring256_bf16");
    return IFH_OK;
}
