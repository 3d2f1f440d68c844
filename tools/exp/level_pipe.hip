// level_pipe.hip -- k_resblock_level (level.hip) with its convolutions SOFTWARE-PIPELINED: no workgroup barrier between the
// eighteen convolutions of a tile, no exposed epilogue except at the end of a residual block.
//
// What level.hip pays per convolution besides its k-steps (tools/probe_level_abl.py, round 5): the barrier (1 360-1 840 clocks of
// wave 0), the last row block's epilogue with nothing to hide under (1 400-1 750), the convolution's set-up -- ~4 000 clocks that
// do not depend on the tap count, 40 / 48 / 59 % of a k = 11 / 7 / 3 convolution.  Both waits have the same cause: every wave has
// to be DONE with convolution c before any wave may start c + 1, because the rows at the edge of a wave's 112-row slab are inputs
// of its neighbours (reach <= 25 rows: taps 11 at dilation 5).  Only the edge: here a wave walks its seven row blocks in the order
//         3, 2, 4 | 1, 5, 0, 6
// The first three read nothing but the wave's own rows 7 .. 104, so they start the moment the wave itself has written them; what
// they need of the previous convolution's last row block (6) is nothing (block 3 reads rows 23 .. 88), so that block's epilogue --
// bias and residual added as the convolution ends, then rounding, LeakyReLU and the image store -- runs under block 3's MFMAs of
// the NEXT convolution ("deferred").  In front of the edge blocks the wave looks at ONE counter in LDS, D, that every wave
// increments once per convolution when all of its rows of the previous one are in the image: D >= 8 n lets convolution n's
// edge blocks read the neighbours' rows -- and overwrite rows the neighbours were reading, and (the same condition) lets the
// weight fragments of convolution n + 2 be sent into the LDS buffer every wave has copied n's out of.  The waves drift apart by at
// most 2/7 of a convolution; nobody waits unless somebody is late.
//
// Image hazards, by construction: convolution c reads image A and writes image B, c + 1 reads B and writes A.  A wave's interior
// epilogues of c + 1 write rows 32 .. 79 of A, which no neighbour reads (they reach rows 0 .. 24 / 87 .. 111 of this wave); its
// edge epilogues come behind the counter wait, when every wave is done reading A for c.  The residual block's last convolution
// (results to `out`, the residual registers refilled with the next block's input) is finished in place as in level.hip.
//
// Arithmetic, k order and rounding points are level.hip's, so the bits are those of three k_resblock_chain launches
// (tests/test_nn_gpu.py::test_resblock_level_is_bit_identical_to_chain_launches runs both forms).
#include "../../infernos_amd/csrc/level.h"

namespace ifh {

// order in which a wave walks its seven row blocks; the first LV_NI read only rows of the wave itself
constexpr int lv_ord(int pos)
{
    constexpr int o[7] = {3, 2, 4, 1, 5, 0, 6};
    return o[pos];
}
constexpr int LV_NI = 3;

// template parameters as k_resblock_level (level.hip); ABL & 16: phase clocks of wave 0 (tools builds)
template <int C, int NW, int WGM, int WGN, int MTB, int HC, int PF, int B0, int B1, int B2, bool ACC0, int ABL = 0>
__global__ __launch_bounds__(NW * 64, 2) void k_resblock_level_pipe(const LevelParams p)
{
    constexpr int NT = 2;
    static_assert(WGM * WGN == NW && WGN * NT * 16 == C, "a workgroup covers every channel");
    static_assert(MTB == 7 && WGN == 1, "the row-block order above is written for seven row blocks of all channels per wave");
    constexpr int RB = C * 2;                          // bytes per image row
    constexpr int SPR = RB / 16;                       // 16-byte slots per row
    constexpr int SH = lv_log2(256 / RB);              // log2(rows per 256-byte bank row)
    constexpr int SWM = SPR / 2 - 1;
    constexpr int RT = WGM * MTB * 16, R = RT - 2 * HC;
    constexpr int GX = 25, GM = 5;                     // guard rows: reach of an 11-tap convolution at dilation 5 / 1
    // the interior row blocks (2, 3, 4) stay inside the wave's rows at the largest reach; the first one (3) does not touch block 6
    static_assert(2 * 16 - GX >= 0 && 4 * 16 + 15 + GX < MTB * 16 && 3 * 16 + 15 + GX < 6 * 16, "row-block order vs reach");
    constexpr int XROWS = RT + 2 * GX, MROWS = RT + 2 * GM;
    constexpr int X_OFF = 0, M_OFF = XROWS * RB;
    constexpr int IMG_BYTES = (XROWS + MROWS) * RB;
    constexpr int KSUB = C / 32, FR = C / 16;
    constexpr int BMAX = B0 > B1 ? (B0 > B2 ? B0 : B2) : (B1 > B2 ? B1 : B2);
    constexpr int KSMAX = BMAX * KSUB;
    constexpr int WB_OFF = IMG_BYTES, WBYTES = KSMAX * FR * 1024;       // two buffers of one convolution's fragments each
    constexpr int D_OFF = WB_OFF + 2 * WBYTES;         // the completion counter
    constexpr int NB = PF + 2;                         // fragment registers: a read lands PF k-steps ahead, its slot was last used 2 steps back
    static_assert(HC >= 6 * (BMAX - 1) || HC == 0, "margin covers the chain: (1+3+5 dilated + 3 plain) * (taps-1)/2 rows");
    static_assert(PF >= 1 && PF <= 3, "prefetch depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int wm = wid % WGM, wn = wid / WGM;
    const int r0 = wm * MTB * 16 + fr;                 // this lane's row of row block 0 (tile-relative)
    const int ch0 = wn * NT * 16 + 4 * g;              // its 4 channels of tile 0 (tile 1: + 16)
    const float slope = p.slope, out_scale = p.out_scale;
    const int T = p.T;

    auto img_addr = [&](int off, int row, int slot) { return off + row * RB + ((slot << 4) ^ (((row >> SH) & SWM) << 5)); };

    for (int i = tid * 16; i < IMG_BYTES; i += NW * 64 * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0, 0, 0, 0);
    if (tid == 0) *reinterpret_cast<uint4 *>(lds + D_OFF) = make_uint4(0, 0, 0, 0);
    __syncthreads();                                   // guard rows stay zero for good

    int xw[NT], mw[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int slot = (wn * NT + i) * 2 + (g >> 1);
        xw[i] = img_addr(X_OFF, GX + r0, slot) + (g & 1) * 8;
        mw[i] = img_addr(M_OFF, GM + r0, slot) + (g & 1) * 8;
    }

    uint2 xr[MTB][NT];                                 // residual stream of the owned elements, packed bf16 (accumulator layout)
    uint2 pvb[3][NT];                                  // rows of `out` to add to (LAST epilogues), loaded two row blocks ahead
    bf16x8_t W[KSMAX][NT];                             // the current convolution's weights for this wave's 32 output channels
#pragma unroll
    for (int s_ = 0; s_ < KSMAX; s_++)
#pragma unroll
        for (int i_ = 0; i_ < NT; i_++) W[s_][i_] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8_t fb[NB];                                   // activation fragments in flight
    f32x4 accs[2][NT];                                 // accumulators of the row block being multiplied / being finished
    f32x4 bv[NT];
    float tv[NT][4];                                   // carried over a convolution's end: its last row block with bias and residual added
    uint2 tpk[NT];
    int rbase[KSMAX];
#pragma unroll
    for (int i_ = 0; i_ < NT; i_++) {
        tv[i_][0] = tv[i_][1] = tv[i_][2] = tv[i_][3] = 0.f;
        tpk[i_] = make_uint2(0, 0);
    }

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    auto srd = [&](const void *base_u) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base_u), 0, 0x7ffffffc, 0x00020000);
    };
    const int lane_col = ch0 * 2;                                        // byte offset of this lane's channels in a row
    auto row_off = [&](int t) __attribute__((always_inline)) { return min(max(t, 0), T - 1) * RB + lane_col; };   // row t clamped into the sequence
    auto ld_row = [&](__amdgpu_buffer_rsrc_t r, int t, int coff) __attribute__((always_inline)) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, row_off(t) + coff, 0, 0);
        return make_uint2(v.x, v.y);
    };
    auto batch_base = [&](const uint16_t *base, int64_t bstride, int tl) { return base + (int64_t)(tl / p.tiles_per_seq) * bstride; };
    auto tile_t0 = [&](int tl) { return (tl % p.tiles_per_seq) * R - HC; };
    // weight fragments of the convolution two ahead: LDS-DMA into one of two buffers (level.hip), issued behind the counter wait
    auto dma_conv = [&](const uint16_t *wbase_u, int soff, int nfr, int par) __attribute__((always_inline)) {
        const unsigned char *g0 = reinterpret_cast<const unsigned char *>(wbase_u) + soff + lane * 16;
#pragma unroll
        for (int jj = 0; jj < (KSMAX * FR + NW - 1) / NW; jj++) {
            const int j = wid + jj * NW;                                 // fragment index: wave-uniform
            if (j < nfr) {
                const unsigned char *g_ = g0 + j * 1024;
                const int dst = WB_OFF + par * WBYTES + j * 1024;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g_), "s"(dst) : "memory");
            }
        }
    };
    const int lane_w = WB_OFF + (wn * NT * 64 + lane) * 16;              // LDS address of this lane's 16 bytes of fragment (s = 0, tile 0), buffer 0
#define LV_WREAD(DSTW, ADDR, OFF)                                                                                    \
    {                                                                                                                \
        const int a_ = (ADDR);                                                                                       \
        bf16x8_t d_;                                                                                                 \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d_) : "v"(a_), "n"(OFF));                                \
        DSTW = d_;                                                                                                   \
    }

    int tile = blockIdx.x;
    if (tile < p.ntiles) {
        const __amdgpu_buffer_rsrc_t xs0 = srd(batch_base(p.x, p.x_bstride, tile));
        const int t00 = tile_t0(tile) + r0;
        lv_static_for<MTB>([&](auto rbc) { lv_static_for<NT>([&](auto ic) { xr[rbc][ic] = ld_row(xs0, t00 + rbc * 16, ic * 32); }); });
        const __amdgpu_buffer_rsrc_t ws0 = srd(p.w[0]);
        lv_static_for<B0 * KSUB>([&](auto sc) {
            lv_static_for<NT>([&](auto ic) {
                const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(ws0, (wn * NT * 64 + lane) * 16, (sc * FR + ic) * 1024, 0);
                W[sc][ic] = __builtin_bit_cast(bf16x8_t, v_);
            });
        });
        dma_conv(p.w[0], B0 * KSUB * FR * 1024, B0 * KSUB * FR, 1);
    }
    int cpar = 0;                                                        // parity of the running convolution count
    unsigned dtarget = 0;                                                // 8 x the convolutions this wave has reported done

#define LV_READ(DST, ADDR, OFF)                                                                                      \
    {                                                                                                                \
        const int a_ = (ADDR);                                                                                       \
        bf16x8_t d_;                                                                                                 \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d_) : "v"(a_), "n"(OFF));                                \
        DST = d_;                                                                                                    \
    }
#define LV_WAIT(N) __builtin_amdgcn_s_waitcnt(0xC07F | ((N) << 8))
#define LV_STORE(ADDR, OFF, VAL)                                                                                     \
    {                                                                                                                \
        const uint2 v_ = (VAL);                                                                                      \
        const int a_ = (ADDR);                                                                                       \
        const unsigned long long q_ = ((unsigned long long)v_.y << 32) | v_.x;                                       \
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a_), "v"(q_), "n"(OFF) : "memory");                       \
    }

    for (; tile < p.ntiles; tile += gridDim.x) {
        const int b = tile / p.tiles_per_seq, ti = tile - b * p.tiles_per_seq;
        const int tq = ti * R - HC;                                      // time of tile row 0
        const bool inside = tq >= 0 && tq + RT <= T;                     // no row of this tile is zero padding
        const int tile_next = tile + (int)gridDim.x < p.ntiles ? tile + (int)gridDim.x : tile;
        const __amdgpu_buffer_rsrc_t outs = srd(p.out + (int64_t)b * p.out_bstride);
        uint32_t okbits = 0, stbits = 0;
#pragma unroll
        for (int rb = 0; rb < MTB; rb++) {
            const int qrow = r0 + rb * 16, t = tq + qrow;
            const bool ok = inside | ((t >= 0) & (t < T));
            okbits |= ok ? (1u << rb) : 0u;
            stbits |= (ok & (qrow >= HC) & (qrow < HC + R)) ? (1u << rb) : 0u;
        }

        // One piece of a row block's epilogue (level.hip), for the row block at position POSP of the order.  KC >= 0: the kind is a
        // compile-time constant; KC == -1: run-time kind; KC == -2: run-time kind that is not 2 (a deferred row block).
        auto piece = [&](auto kc_c, auto acc_c, auto posp_c, auto p_c, int kind, int tqo, int wdst0, int wdst1, __amdgpu_buffer_rsrc_t xnext, int tnext)
                         __attribute__((always_inline)) {
            constexpr int KC = decltype(kc_c)::value, POSP = decltype(posp_c)::value, RBP = lv_ord(POSP), P = decltype(p_c)::value;
            constexpr bool ACC = decltype(acc_c)::value;
            constexpr int i = P / 4, Q = P % 4;
            constexpr bool NOT2 = KC == -2;
            const int kd = KC >= 0 ? KC : kind;
#define LV_BRANCH() if constexpr (KC < 0) asm volatile("" ::: "memory")
            if constexpr (Q == 0) {
                const f32x4 a = accs[POSP & 1][i];
                tv[i][0] = a[0] + bv[i][0]; tv[i][1] = a[1] + bv[i][1]; tv[i][2] = a[2] + bv[i][2]; tv[i][3] = a[3] + bv[i][3];
                if (kd != 0) {
                    LV_BRANCH();
                    const uint2 rv = xr[RBP][i];
                    tv[i][0] += __uint_as_float(rv.x << 16);
                    tv[i][1] += __uint_as_float(rv.x & 0xffff0000u);
                    tv[i][2] += __uint_as_float(rv.y << 16);
                    tv[i][3] += __uint_as_float(rv.y & 0xffff0000u);
                }
            } else if constexpr (Q == 1) {
                if constexpr (!NOT2) {
                    if (kd == 2) {
                        LV_BRANCH();
                        tv[i][0] *= out_scale; tv[i][1] *= out_scale; tv[i][2] *= out_scale; tv[i][3] *= out_scale;
                        if constexpr (ACC) {
                            const uint2 q2 = pvb[POSP % 3][i];
                            tv[i][0] += __uint_as_float(q2.x << 16);
                            tv[i][1] += __uint_as_float(q2.x & 0xffff0000u);
                            tv[i][2] += __uint_as_float(q2.y << 16);
                            tv[i][3] += __uint_as_float(q2.y & 0xffff0000u);
                        }
                    }
                }
                tpk[i] = make_uint2(f32x2_to_bf16x2(tv[i][0], tv[i][1]), f32x2_to_bf16x2(tv[i][2], tv[i][3]));
                if (kd == 1) {
                    LV_BRANCH();
                    xr[RBP][i] = tpk[i];
                }
            } else if constexpr (Q == 2) {
                if (NOT2 || kd != 2) {
                    if constexpr (!NOT2) LV_BRANCH();
                    tpk[i] = lv_lrelu4(tpk[i], slope);
                }
            } else {
                if (NOT2 || kd != 2) {
                    const uint32_t okm = (uint32_t)((int32_t)(okbits << (31 - RBP)) >> 31);      // all ones / zero: one v_bfe_i32
                    LV_STORE((i == 0 ? wdst0 : wdst1), RBP * 16 * RB, make_uint2(tpk[i].x & okm, tpk[i].y & okm))
                } else {
                    if ((stbits >> RBP) & 1u)
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){tpk[i].x, tpk[i].y}, outs, (tqo + r0 + RBP * 16) * RB + lane_col + i * 32, 0, 0);
                    xr[RBP][i] = ld_row(xnext, tnext + RBP * 16, i * 32);
                }
            }
#undef LV_BRANCH
        };

        // Convolution q of a block (kinds as in level.hip: 0 dilated x image -> intermediate image, 1 plain + residual -> x image,
        // 2 the block's last one -> out).  kprev: kind of the convolution in front of it whose last row block (6) is still in tv, to be
        // finished under this one's first row block and stored into THIS convolution's source image (pdst0/1); -1: nothing pending.
        auto conv = [&](auto taps_c, auto acc_c, auto nks_c, int kind, int kprev, int src_off, int G, int d, int wdst0, int wdst1, int pdst0, int pdst1,
                        __amdgpu_buffer_rsrc_t bsrd, int boff, const uint16_t *w2base, int w2off, int w2nfr, __amdgpu_buffer_rsrc_t xnext, int tnext)
                        __attribute__((always_inline)) {
            constexpr int TAPS = decltype(taps_c)::value, NKS = decltype(nks_c)::value;
            constexpr bool ACC = decltype(acc_c)::value;
            constexpr int KS = TAPS * KSUB, H = (TAPS - 1) / 2;
            constexpr int N1 = LV_NI * KS, N2 = (MTB - LV_NI) * KS;      // k-steps of the interior / edge row blocks
            unsigned long long tprev = (ABL & 16) ? __builtin_amdgcn_s_memtime() : 0;
#define LV_STAMP(IDX)                                                      \
    if constexpr ((ABL & 16) != 0) {                                       \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        pf[IDX] += now_ - tprev;                                           \
        tprev = now_;                                                      \
    }
            int tqo = __builtin_amdgcn_readfirstlane(tq), dd = __builtin_amdgcn_readfirstlane(d);
            asm volatile("" : "+s"(tqo), "+s"(dd));
            asm volatile("" : "+v"(tnext));
            d = dd;
            // this wave's pieces of the DMA issued in the previous convolution have landed (reported with this convolution's count)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lv_static_for<NT>([&](auto ic) { bv[ic] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bsrd, ch0 * 4 + ic * 64, boff, 0)); });
            lv_static_for<KS>([&](auto sc) {
                constexpr int s = decltype(sc)::value, tap = s / KSUB, cs = s % KSUB;
                rbase[s] = img_addr(src_off, G + r0 + (tap - H) * d, cs * 4 + g);
            });
            if constexpr (ACC)                         // rows of `out` for the first row block (the others: one row block before their block)
                if (kind == 2) lv_static_for<NT>([&](auto ic) { pvb[0][ic] = ld_row(outs, tqo + r0 + lv_ord(0) * 16, ic * 32); });
            const int wrd = lane_w + ((cpar + 1) & 1) * WBYTES;          // where the next convolution's fragments are read from
            LV_STAMP(0)                                // [0] set-up
            __builtin_amdgcn_sched_barrier(0);
            // k-step s of the row block at position pos; PH: 0 = interior positions 0 .. LV_NI-1, 1 = the rest (their fragment reads
            // form two separate pipelines: nothing of the edge blocks is read in front of the counter wait)
            auto step = [&](auto kc_c, auto ph_c, auto pos_c, auto sc) __attribute__((always_inline)) {
                constexpr int KC = decltype(kc_c)::value, PH = decltype(ph_c)::value, pos = decltype(pos_c)::value, s = decltype(sc)::value;
                constexpr int base = PH == 0 ? 0 : LV_NI, NPH = PH == 0 ? N1 : N2;
                constexpr int n = (pos - base) * KS + s, n2 = n + PF;
                const int kd = KC >= 0 ? KC : kind;
                if constexpr (n2 < NPH) LV_READ(fb[n2 % NB], rbase[n2 % KS], lv_ord(base + n2 / KS) * 16 * RB);
                LV_WAIT(lv_min(PF, NPH - 1 - n));
                __builtin_amdgcn_sched_barrier(0);
#ifdef LVP_PRIO
                __builtin_amdgcn_s_setprio(LVP_PRIO);
#endif
                lv_static_for<NT>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (s == 0)
                        accs[pos & 1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[s][i], fb[n % NB], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else
                        accs[pos & 1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[s][i], fb[n % NB], accs[pos & 1][i], 0, 0, 0);
                });
#ifdef LVP_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
#ifndef LVP_NOSB_AFTER
                __builtin_amdgcn_sched_barrier(0);
#endif
                if constexpr (pos == MTB - 1) {        // the next convolution's fragments out of LDS, as this one frees the registers
                    lv_static_for<NT>([&](auto ic) { LV_WREAD(W[s][ic], wrd, (s * FR + ic) * 1024) });
                    if constexpr (s == KS - 1 && NKS > KS)
                        if (kind == 2)
                            lv_static_for<NKS - KS>([&](auto s2c) { lv_static_for<NT>([&](auto ic) { LV_WREAD(W[KS + s2c][ic], wrd, ((KS + s2c) * FR + ic) * 1024) }); });
                }
                if constexpr (ACC && s == 0 && pos + 1 < MTB)
                    if (kd == 2) lv_static_for<NT>([&](auto ic) { pvb[(pos + 1) % 3][ic] = ld_row(outs, tqo + r0 + lv_ord(pos + 1) * 16, ic * 32); });
                if constexpr (pos > 0) {
                    constexpr int NP = 4 * NT;
                    lv_static_for<NP>([&](auto pc) {
                        if constexpr (1 + (decltype(pc)::value * (KS - 1)) / NP == s)
                            piece(kc_c, acc_c, std::integral_constant<int, pos - 1>{}, pc, kind, tqo, wdst0, wdst1, xnext, tnext);
                    });
                } else {
                    // the previous convolution's last row block: rounding, LeakyReLU and store of tile i at step 1 + i (KS >= 3)
                    constexpr std::integral_constant<int, -2> not2{};
                    constexpr std::integral_constant<int, MTB - 1> lastpos{};
                    lv_static_for<NT>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        if constexpr (s == 1 + i) {
                            if (kprev >= 0) {
                                asm volatile("" ::: "memory");
                                piece(not2, acc_c, lastpos, std::integral_constant<int, i * 4 + 1>{}, kprev, tqo, pdst0, pdst1, xnext, tnext);
                                piece(not2, acc_c, lastpos, std::integral_constant<int, i * 4 + 2>{}, kprev, tqo, pdst0, pdst1, xnext, tnext);
                                piece(not2, acc_c, lastpos, std::integral_constant<int, i * 4 + 3>{}, kprev, tqo, pdst0, pdst1, xnext, tnext);
                            }
                        }
                    });
                    // every row of the previous convolution (or of the block's x image) is in LDS: report it
                    if constexpr (s == KS - 1) {
                        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(D_OFF), "v"(1u) : "memory");
                        dtarget += 8;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto drain = [&]() __attribute__((always_inline)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                lv_static_for<NB>([&](auto jc) { bf16x8_t d_ = fb[jc]; asm volatile("" : "+v"(d_)); fb[jc] = d_; });
                __builtin_amdgcn_sched_barrier(0);
            };
            // Whatever a path reads from LDS by inline asm it reads AND waits for inside the path (level.hip: at a join hipcc may copy
            // a register that an asm load has not delivered yet)
            auto interior = [&](auto kc_c) __attribute__((always_inline)) {
                lv_static_for<PF>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    LV_READ(fb[n % NB], rbase[n % KS], lv_ord(n / KS) * 16 * RB);
                });
                __builtin_amdgcn_sched_barrier(0);
                lv_static_for<LV_NI>([&](auto posc) { lv_static_for<KS>([&](auto sc) { step(kc_c, std::integral_constant<int, 0>{}, posc, sc); }); });
                drain();
            };
            auto edge = [&](auto kc_c) __attribute__((always_inline)) {
                lv_static_for<PF>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    LV_READ(fb[n % NB], rbase[n % KS], lv_ord(LV_NI + n / KS) * 16 * RB);
                });
                __builtin_amdgcn_sched_barrier(0);
                lv_static_for<MTB - 1 - LV_NI>([&](auto pc) {
                    lv_static_for<KS>([&](auto sc) { step(kc_c, std::integral_constant<int, 1>{}, std::integral_constant<int, LV_NI + decltype(pc)::value>{}, sc); });
                });
                drain();
            };
            if (kind == 0) interior(std::integral_constant<int, 0>{});
            else if (kind == 1) interior(std::integral_constant<int, 1>{});
            else interior(std::integral_constant<int, 2>{});
            LV_STAMP(1)                                // [1] interior row blocks (+ the deferred epilogue)
            // every wave has reported the previous convolution: its rows may be read, the rows it read overwritten, the fragment
            // buffer it was copied out of refilled
            {
                unsigned seen;
                do {
                    unsigned v_;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v_) : "v"(D_OFF) : "memory");
                    seen = __builtin_amdgcn_readfirstlane(v_);
                    if ((int)(seen - dtarget) < 0) __builtin_amdgcn_s_sleep(1);
                } while ((int)(seen - dtarget) < 0);
            }
            LV_STAMP(2)                                // [2] counter wait
            dma_conv(w2base, w2off, w2nfr, cpar & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (kind == 0) edge(std::integral_constant<int, 0>{});
            else if (kind == 1) edge(std::integral_constant<int, 1>{});
            else edge(std::integral_constant<int, 2>{});
            LV_STAMP(3)                                // [3] edge row blocks but the last
            constexpr std::integral_constant<int, -1> any_kind{};
            lv_static_for<KS>([&](auto sc) { step(any_kind, std::integral_constant<int, 1>{}, std::integral_constant<int, MTB - 1>{}, sc); });
            LV_STAMP(4)                                // [4] last row block (+ fragment copies)
            // the next convolution's fragments are in W before this body is left (the loop's back edge is a join too)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lv_static_for<KSMAX>([&](auto sc) { lv_static_for<NT>([&](auto ic) { bf16x8_t d_ = W[sc][ic]; asm volatile("" : "+v"(d_)); W[sc][ic] = d_; }); });
            __builtin_amdgcn_sched_barrier(0);
            constexpr std::integral_constant<int, MTB - 1> lastpos{};
            if (kind == 2) {                           // a block's last convolution is finished here: the next block starts from its refills
                asm volatile("" ::: "memory");
                lv_static_for<4 * NT>([&](auto pc) { piece(any_kind, acc_c, lastpos, pc, kind, tqo, wdst0, wdst1, xnext, tnext); });
            } else {                                   // bias (+ residual) now, the rest under the next convolution's first row block
                constexpr std::integral_constant<int, -2> not2{};
                piece(not2, acc_c, lastpos, std::integral_constant<int, 0>{}, kind, tqo, wdst0, wdst1, xnext, tnext);
                piece(not2, acc_c, lastpos, std::integral_constant<int, 4>{}, kind, tqo, wdst0, wdst1, xnext, tnext);
            }
            __builtin_amdgcn_sched_barrier(0);
            cpar ^= 1;
            LV_STAMP(5)                                // [5] the convolution's end
            if constexpr ((ABL & 16) != 0) pf[6] += 1;                 // [6] convolutions
#undef LV_STAMP
        };

        auto block = [&](auto taps_c, auto nks_c, auto acc_c, const uint16_t *wb, const float *bb, const uint16_t *wnext_block,
                         int tile_nx) __attribute__((always_inline)) {
            constexpr int TAPS = decltype(taps_c)::value, KS = TAPS * KSUB, NKS = decltype(nks_c)::value;
            const unsigned long long tblk = (ABL & 16) ? __builtin_amdgcn_s_memtime() : 0;
            // x image = LeakyReLU(x), zero outside the sequence.  Every wave is past the convolution that read the previous x image:
            // this wave's counter wait in the block's last convolution (or the zero fill's barrier) saw them report it.
            lv_static_for<MTB>([&](auto rbc) {
                constexpr int rb = decltype(rbc)::value;
                const uint32_t okm = (uint32_t)((int32_t)(okbits << (31 - rb)) >> 31);
                lv_static_for<NT>([&](auto ic) {
                    const uint2 v = lv_lrelu4(xr[rb][ic], slope);
                    LV_STORE(xw[ic], rb * 16 * RB, make_uint2(v.x & okm, v.y & okm))
                });
            });
            if constexpr ((ABL & 16) != 0) {
                const unsigned long long now_ = __builtin_amdgcn_s_memtime();
                pf[7] += now_ - tblk;                                     // [7] block top (x image fill)
            }
            const __amdgpu_buffer_rsrc_t xnext = srd(batch_base(p.x, p.x_bstride, tile_nx));
            const int tnext = tile_t0(tile_nx) + r0;
            const __amdgpu_buffer_rsrc_t bsb = srd(bb);
            constexpr int CB = KS * FR * 1024;                            // bytes per convolution in the fragment stream
#pragma unroll 1
            for (int q = 0; q < 6; q++) {
                const bool dil = (q & 1) == 0, last = q == 5, own2 = q < 4;      // own2: the convolution two ahead is this block's
                conv(taps_c, acc_c, nks_c, last ? 2 : (q & 1), q == 0 ? -1 : ((q - 1) & 1), dil ? X_OFF : M_OFF, dil ? GX : GM, dil ? q + 1 : 1,
                     dil ? mw[0] : xw[0], dil ? mw[1] : xw[1], dil ? xw[0] : mw[0], dil ? xw[1] : mw[1],
                     bsb, q * C * 4, own2 ? wb : wnext_block, own2 ? (q + 2) * CB : (q - 4) * NKS * FR * 1024, own2 ? KS * FR : NKS * FR, xnext, tnext);
            }
        };

        constexpr int NBLK = (B0 > 0) + (B1 > 0) + (B2 > 0);
        block(std::integral_constant<int, B0>{}, std::integral_constant<int, (NBLK > 1 ? B1 : B0) * KSUB>{}, std::integral_constant<bool, ACC0>{}, p.w[0], p.bias[0],
              NBLK > 1 ? p.w[1] : p.w[0], NBLK > 1 ? tile : tile_next);
        if constexpr (B1 > 0)
            block(std::integral_constant<int, B1>{}, std::integral_constant<int, (NBLK > 2 ? B2 : B0) * KSUB>{}, std::true_type{}, p.w[1], p.bias[1],
                  NBLK > 2 ? p.w[2] : p.w[0], NBLK > 2 ? tile : tile_next);
        if constexpr (B2 > 0)
            block(std::integral_constant<int, B2>{}, std::integral_constant<int, B0 * KSUB>{}, std::true_type{}, p.w[2], p.bias[2], p.w[0], tile_next);
    }
    if constexpr ((ABL & 16) != 0) {
        if (tid == 0)
            for (int i = 0; i < 8; i++) atomicAdd(p.prof + i, pf[i]);
    }
#undef LV_READ
#undef LV_WREAD
#undef LV_WAIT
#undef LV_STORE
}

template <int C, int NW, int WGM, int WGN, int MTB, int HC, int PF, int B0, int B1, int B2, bool ACC0, int ABL = 0>
static int launch_pipe(LevelParams &p, hipStream_t st)
{
    constexpr int RT = WGM * MTB * 16, R = RT - 2 * HC;
    constexpr int BMAX = B0 > B1 ? (B0 > B2 ? B0 : B2) : (B1 > B2 ? B1 : B2);
    constexpr size_t bytes = (size_t)(RT + 50 + RT + 10) * C * 2 + 2 * (size_t)BMAX * (C / 32) * (C / 16) * 1024 + 16;
    static_assert(bytes <= 160 * 1024, "tile does not fit in LDS");
    auto kern = k_resblock_level_pipe<C, NW, WGM, WGN, MTB, HC, PF, B0, B1, B2, ACC0, ABL>;
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return check_hip(e, "resblock_level_pipe lds attr");
        attr_once.done(attr_dev);
    }
    p.tiles_per_seq = (p.T + R - 1) / R;
    p.ntiles = p.tiles_per_seq * p.nbatch;
    const int ncu = device_cu_count();
    if (ncu <= 0) return fail(IFH_EHIP, "resblock_level: device query");
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), bytes, st, p);
    return IFH_OK;
}

int launch_level_pipe(LevelParams &p, int t0, int t1, int t2, hipStream_t st)
{
    int rc = IFH_EINVAL;
#define PIPE_CASE(A, B, C_)                                                                                     \
    if (t0 == A && t1 == B && t2 == C_)                                                                         \
        rc = p.accumulate ? launch_pipe<32, 8, 8, 1, 7, 64, 3, A, B, C_, true>(p, st) : launch_pipe<32, 8, 8, 1, 7, 64, 3, A, B, C_, false>(p, st);
#ifdef LV_DEV_ABL          /* tools builds: phase clocks (IFH_LEVEL_ABL=16) */
    const int abl = getenv("IFH_LEVEL_ABL") ? atoi(getenv("IFH_LEVEL_ABL")) : 0;
    if (abl == 16 && t1 == 0 && t0 == 3) return launch_pipe<32, 8, 8, 1, 7, 64, 3, 3, 0, 0, false, 16>(p, st);
    if (abl == 16 && t1 == 0 && t0 == 7) return launch_pipe<32, 8, 8, 1, 7, 64, 3, 7, 0, 0, false, 16>(p, st);
    if (abl == 16 && t1 == 0 && t0 == 11) return launch_pipe<32, 8, 8, 1, 7, 64, 3, 11, 0, 0, false, 16>(p, st);
#endif
#ifdef LV_DEV_ONLY
    PIPE_CASE(LV_DEV_ONLY, 0, 0)
#else
    PIPE_CASE(3, 7, 11)
    PIPE_CASE(3, 0, 0)
    PIPE_CASE(7, 0, 0)
    PIPE_CASE(11, 0, 0)
#endif
#undef PIPE_CASE
    return rc;
}

}  // namespace ifh
