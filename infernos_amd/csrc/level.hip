// level.hip -- the residual blocks of one HiFi-GAN upsampling level in ONE launch, weights stationary in registers:
//     for each block j (taps 3, 7, 11):  y = x;  for d in (1, 3, 5):  y = y + conv_k,1(lrelu(conv_k,d(lrelu(y))))
//     out = y * out_scale (+ out)              -- with out_scale = 1/3 over the three blocks: the level's mean
// (transformers modeling_speecht5.py HifiGanResidualBlock.forward / SpeechT5HifiGan.forward; reached from
// HelloSippyTTSRT/HelloSippyRTPipe.py:236).  Same arithmetic, k order and rounding points as k_resblock_chain
// (chain.hip), one launch per block with `accumulate` -- bit-identical to it -- but a different machine:
//
//   * k_resblock_chain reads BOTH MFMA operands from LDS (weights through a DMA ring, 179 B/clk of fragment reads at full
//     matrix rate at C = 32), runs k-step-major over all of a wave's tiles, so its six epilogues per tile are serial phases
//     of all eight waves (28-37 % of a tile at C <= 64), and meets at a barrier per 8 KB weight unit.
//   * Here a wave keeps the WHOLE convolution's weights for its 32 output channels in registers (A operand: 88 VGPRs at
//     C = 32, k = 11; loaded from L2/L1 as pre-packed fragments, reloaded fragment by fragment during the last row block of
//     the previous convolution) and walks its rows ROW-BLOCK-major: 16 rows x 32 channels per block, one 1 KB activation
//     fragment (B operand, ds_read_b128) per two MFMAs = 128 B/clk at full rate.  The epilogue of row block r (bias,
//     residual, rounding, LeakyReLU, image store) is cut into eight pieces that are issued between the MFMAs of row block
//     r + 1, so the vector work runs in the matrix pipe's shadow; only the last block's epilogue of a convolution is exposed.
//     One barrier per convolution.
//   * The operand images are unpadded (C * 2 bytes per row) with an XOR swizzle of the 16-byte slots,
//     slot ^= ((row >> log2(rows per 256 B)) & (slots/2 - 1)) << 1, conflict-free for the fragment read whatever row
//     it starts at: the tile is 896 rows at C = 32 (768 stored: 14 % recomputed margin instead of 20 %).
//   * All blocks of the level run back to back on the same tile: x is read once per block from L2 (prefetched into the
//     residual registers as the previous block's last epilogues free them), the running mean goes through `out` (L2).
#include "level.h"

namespace ifh {

// C channels; NW waves = WGM row groups x WGN channel groups, a wave owns MTB row blocks of 16 rows x 32 channels (two 16 x 16
// tiles); HC rows computed on either side of the R = WGM*MTB*16 - 2*HC stored rows; PF = fragment reads in flight;
// B0, B1, B2 = taps of the blocks run back to back (0 = none); ACC0: the first block adds to `out` too;
// ABL (tools builds only, wrong results): 1 = no fragment reloads, 2 = no epilogue pieces, 4 = no activation reads, 8 = no MFMAs
// POST: the 7-tap 32 -> 1 channel convolution + tanh behind the level (SpeechT5HifiGan.forward's conv_post, what k_conv_post of
// misc.hip computes, in its order of sums) runs on the tile's final mean while it is still in LDS: the level's output never crosses
// HBM, the running mean of the blocks lives in a workspace slab of the workgroup's own (L2-resident), the kernel writes audio.
template <int C, int NW, int WGM, int WGN, int MTB, int HC, int PF, int B0, int B1, int B2, bool ACC0, int ABL = 0, bool POST = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void k_resblock_level(const LevelParams p)
{
    constexpr int NT = 2;
    static_assert(WGM * WGN == NW && WGN * NT * 16 == C, "a workgroup covers every channel");
    constexpr int RB = C * 2;                          // bytes per image row
    constexpr int SPR = RB / 16;                       // 16-byte slots per row
    constexpr int SH = lv_log2(256 / RB);              // log2(rows per 256-byte bank row)
    constexpr int SWM = SPR / 2 - 1;
    constexpr int RT = WGM * MTB * 16, R = RT - 2 * HC;
    constexpr int GX = 25, GM = 5;                     // guard rows: reach of an 11-tap convolution at dilation 5 / 1
    constexpr int XROWS = RT + 2 * GX, MROWS = RT + 2 * GM;
    constexpr int X_OFF = 0, M_OFF = XROWS * RB;
    constexpr int IMG_BYTES = (XROWS + MROWS) * RB;
    constexpr int KSUB = C / 32, FR = C / 16;
    constexpr int BMAX = B0 > B1 ? (B0 > B2 ? B0 : B2) : (B1 > B2 ? B1 : B2);
    constexpr int KSMAX = BMAX * KSUB;
    constexpr int WB_OFF = IMG_BYTES, WBYTES = KSMAX * FR * 1024;       // two buffers of one convolution's fragments each
    constexpr int NB = PF + 2;                         // fragment registers: a read lands PF k-steps ahead, its slot was last used 2 steps back
    constexpr int NP = 4 * NT;                         // epilogue pieces per row block
    static_assert(HC >= 6 * (BMAX - 1) + (POST ? 3 : 0) || HC == 0, "margin covers the chain: (1+3+5 dilated + 3 plain) * (taps-1)/2 rows (+ conv_post's 3)");
    static_assert(!POST || (C == 32 && B2 > 0 && !ACC0), "the folded conv_post is the C = 32 level's: three blocks, nothing to add to");
    static_assert(PF >= 1 && PF <= 6, "prefetch depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    // ABL & 16: phase clocks of wave 0, summed locally and added to p.prof once as the workgroup ends (an atomic per stamp -- a vector
    // memory operation in front of the kernel's own vmcnt waits -- tripled the kernel's time and put most of it into the W-wait stamp)
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
    __syncthreads();                                   // guard rows stay zero for good

    // store bases of this lane (row block 0): 8 bytes = its 4 channels of tile i in row GX/GM + r0
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
    float tv[NT][4];
    uint2 tpk[NT];
    int rbase[KSMAX];

    // Global memory goes through buffer instructions: a wave-uniform descriptor (4 SGPRs) + a 32-bit per-lane byte offset + an
    // SGPR offset.  (As 64-bit pointers hipcc hoisted every row and fragment address of a tile out of its 18 convolutions -- more
    // than 100 live VGPRs, spills.)
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
    // The weights reach the registers through LDS: fragment (s, tile) of the convolution two ahead is DMA'd (global_load_lds, 1 KB per
    // wave-instruction = one fragment, no VGPR round trip) into one of two buffers as a convolution starts -- a whole convolution
    // before the barrier that publishes it -- and every wave copies its 2 * KS fragments of the NEXT convolution out of the other
    // buffer into W during its last row block, as the MFMAs free them.  (Straight from L2 each wave's 22 KB of k = 11 fragments were
    // requested in the last row block and not back before the next convolution's first MFMA: ~10 k clocks per convolution.)
    // The DMA is inline asm so that hipcc does not count it (it would drain vmcnt to 0 at every use of a load of its own).
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
        // convolution 0's fragments straight from memory, convolution 1's into buffer 1 (convolution c reads buffer (c + 1) & 1 ...)
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

#define LV_READ(DST, ADDR, OFF)                                                                                      \
    {                                                                                                                \
        const int a_ = (ADDR);                                                                                       \
        bf16x8_t d_; /* (clang: an asm operand inside a nested generic lambda cannot name a captured variable) */     \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d_) : "v"(a_), "n"(OFF));                                \
        DST = d_;                                                                                                    \
    }
// s_waitcnt lgkmcnt(N) alone (vmcnt / expcnt fields at "no wait"); a builtin, so no inline-asm boundary in front of the MFMAs
// (the tied-operand asm form cost an s_nop per k-step); the sched_barrier behind it keeps the MFMAs below it
#define LV_WAIT(FRAG, N) __builtin_amdgcn_s_waitcnt(0xC07F | ((N) << 8))
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
        // POST: the running mean of this tile's rows in the workgroup's own slab (every row of the tile, margins included: the last
        // block needs the mean three rows beyond the stored ones); else in `out`
        const __amdgpu_buffer_rsrc_t outs = POST ? srd(p.mean_ws + (size_t)blockIdx.x * RT * C) : srd(p.out + (int64_t)b * p.out_bstride);
        auto ld_prev = [&](int tqo_, int rb, int coff) __attribute__((always_inline)) {          // rows of the running mean for row block rb
            if constexpr (POST) {
                // (tqo_ carries this lane's slab offset here: an opaque per-convolution copy, or hipcc keeps one address register per
                // row block and tile alive through all eighteen convolutions; the row block's part goes into the scalar offset)
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(outs, tqo_, rb * 16 * RB + coff, 0);
                return make_uint2(v.x, v.y);
            } else
                return ld_row(outs, tqo_ + r0 + rb * 16, coff);
        };
        // per lane: bit rb = its row of row block rb lies inside the sequence (okbits) / is one of the tile's stored rows too (stbits)
        uint32_t okbits = 0, stbits = 0;
#pragma unroll
        for (int rb = 0; rb < MTB; rb++) {
            const int qrow = r0 + rb * 16, t = tq + qrow;
            const bool ok = inside | ((t >= 0) & (t < T));
            okbits |= ok ? (1u << rb) : 0u;
            stbits |= (ok & (qrow >= HC) & (qrow < HC + R)) ? (1u << rb) : 0u;
        }

        // The six convolutions of a block run through ONE loop body (kind is wave-uniform run-time data): the fragments W are then a
        // loop-carried value that the last row block refills in place.  (As separate bodies for the dilated / plain / last
        // convolution hipcc gave each body's W its own registers -- two full sets alive, spills at k = 11 -- and the code was 3 x larger.)
        // kind 0: dilated convolution, x image -> intermediate image.  kind 1: plain convolution + residual -> new residual
        // (registers) and its LeakyReLU'd x image.  kind 2: the block's last convolution: + residual, * out_scale (+ out) -> out,
        // and the residual registers are refilled with the raw rows the NEXT block (or tile) starts from.
        // KC >= 0: the kind is a compile-time constant (row blocks 0 .. MTB-2 of a convolution: three instantiations chosen by one
        // branch per convolution); KC < 0: run-time kind (the last row block, which also refills W -- one body for all kinds).
        auto piece = [&](auto kc_c, auto acc_c, auto last_c, auto rbp_c, auto p_c, int kind, int tqo, int wso, int wdst0, int wdst1, __amdgpu_buffer_rsrc_t xnext, int tnext)
                         __attribute__((always_inline)) {
            constexpr int KC = decltype(kc_c)::value, RBP = decltype(rbp_c)::value, P = decltype(p_c)::value;
            constexpr bool ACC = decltype(acc_c)::value;
            constexpr int i = P / 4, Q = P % 4;
            const int kd = KC >= 0 ? KC : kind;
            // with a run-time kind the kind-dependent parts are real (wave-uniform) branches: the empty asm keeps hipcc from turning
            // them into both-sides-plus-select code (95 VALU per row block instead of ~45)
#define LV_BRANCH() if constexpr (KC < 0) asm volatile("" ::: "memory")
            if constexpr (Q == 0) {
                const f32x4 a = accs[RBP & 1][i];
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
                if (kd == 2) {
                    LV_BRANCH();
                    tv[i][0] *= out_scale; tv[i][1] *= out_scale; tv[i][2] *= out_scale; tv[i][3] *= out_scale;
                    if constexpr (ACC) {
                        const uint2 q2 = pvb[RBP % 3][i];
                        tv[i][0] += __uint_as_float(q2.x << 16);
                        tv[i][1] += __uint_as_float(q2.x & 0xffff0000u);
                        tv[i][2] += __uint_as_float(q2.y << 16);
                        tv[i][3] += __uint_as_float(q2.y & 0xffff0000u);
                    }
                }
                tpk[i] = make_uint2(f32x2_to_bf16x2(tv[i][0], tv[i][1]), f32x2_to_bf16x2(tv[i][2], tv[i][3]));
                if (kd == 1) {
                    LV_BRANCH();
                    xr[RBP][i] = tpk[i];
                }
            } else if constexpr (Q == 2) {
                if (kd != 2) {
                    LV_BRANCH();
                    tpk[i] = lv_lrelu4(tpk[i], slope);
                }
            } else {
                if (kd != 2) {
                    const uint32_t okm = (uint32_t)((int32_t)(okbits << (31 - RBP)) >> 31);      // all ones / zero: one v_bfe_i32
                    LV_STORE((i == 0 ? wdst0 : wdst1), RBP * 16 * RB, make_uint2(tpk[i].x & okm, tpk[i].y & okm))
                } else {
                    if constexpr (POST) {
                        if constexpr (decltype(last_c)::value)             // the level's mean of this row: into the (free) x image, for the conv_post phase
                            LV_STORE((i == 0 ? xw[0] : xw[1]), RBP * 16 * RB, tpk[i])
                        else                                               // the running mean: every row of the tile, to the workgroup's slab
                            __builtin_amdgcn_raw_buffer_store_b64((u32x2){tpk[i].x, tpk[i].y}, outs, wso, RBP * 16 * RB + i * 32, 0);
                    } else {
                        if ((stbits >> RBP) & 1u)
                            __builtin_amdgcn_raw_buffer_store_b64((u32x2){tpk[i].x, tpk[i].y}, outs, (tqo + r0 + RBP * 16) * RB + lane_col + i * 32, 0, 0);
                    }
                    xr[RBP][i] = ld_row(xnext, tnext + RBP * 16, i * 32);
                }
            }
#undef LV_BRANCH
        };

        // Convolution q of a block: over the image at src_off (guard G rows) with dilation d, results to the image behind wdst0/1
        // (kinds 0, 1).  (w2base, w2off, w2nfr): the convolution TWO ahead, whose fragments this one's start sends on their way to LDS;
        // NKS: k-steps of the convolution that follows a block's last one (kind 2); xnext/tnext: kind 2's refill of the residual registers.
        auto conv = [&](auto taps_c, auto acc_c, auto nks_c, auto last_c, int kind, int src_off, int G, int d, int wdst0, int wdst1, __amdgpu_buffer_rsrc_t bsrd,
                        int boff, const uint16_t *w2base, int w2off, int w2nfr, __amdgpu_buffer_rsrc_t xnext, int tnext) __attribute__((always_inline)) {
            constexpr int TAPS = decltype(taps_c)::value, NKS = decltype(nks_c)::value;
            constexpr bool ACC = decltype(acc_c)::value;
            constexpr int KS = TAPS * KSUB, H = (TAPS - 1) / 2, N = MTB * KS;
            unsigned long long tprev = (ABL & 16) ? __builtin_amdgcn_s_memtime() : 0;
#define LV_STAMP(IDX)                                                      \
    if constexpr ((ABL & 16) != 0) {                                       \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        pf[IDX] += now_ - tprev;                                           \
        tprev = now_;                                                      \
    }
            // what this convolution derives from (tq, d, tnext) is derived HERE: opaque copies keep hipcc from hoisting row addresses
            // and fragment bases out of the loop over the convolutions
            int tqo = __builtin_amdgcn_readfirstlane(tq), dd = __builtin_amdgcn_readfirstlane(d);
            asm volatile("" : "+s"(tqo), "+s"(dd));
            asm volatile("" : "+v"(tnext));
            int wso = r0 * RB + lane_col;              // POST: this lane's byte offset in the workgroup's slab (row block 0), opaque per convolution
            asm volatile("" : "+v"(wso));
            d = dd;
            // This wave's pieces of the DMA issued as the previous convolution started have landed (nothing else of its vector memory
            // traffic is pending here: dilated / plain convolutions have none, and a block's top drains what its predecessor's last one left)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LV_STAMP(2)                                // [2] fragment DMA of the previous convolution landed
            lv_static_for<NT>([&](auto ic) { bv[ic] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bsrd, ch0 * 4 + ic * 64, boff, 0)); });
            lv_static_for<KS>([&](auto sc) {
                constexpr int s = decltype(sc)::value, tap = s / KSUB, cs = s % KSUB;
                rbase[s] = img_addr(src_off, G + r0 + (tap - H) * d, cs * 4 + g);
            });
            if constexpr (ACC)                         // rows of `out` for row block 0 (the others: one row block before their block)
                if (kind == 2) lv_static_for<NT>([&](auto ic) { pvb[0][ic] = ld_prev(POST ? wso : tqo, 0, ic * 32); });
            const int wrd = lane_w + ((cpar + 1) & 1) * WBYTES;          // where the next convolution's fragments are read from
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            LV_STAMP(0)                                // [0] set-up
            __builtin_amdgcn_s_barrier();              // every wave has written its rows of the source image, every piece of the DMA is in LDS
            LV_STAMP(1)                                // [1] barrier
            if constexpr (!(ABL & 1)) dma_conv(w2base, w2off, w2nfr, cpar & 1);
            __builtin_amdgcn_sched_barrier(0);
            // k-step s of row block rb: the activation fragment PF steps ahead is requested, this step's two MFMAs are issued, and
            // the pieces of row block rb - 1's epilogue that belong to this step follow them
            auto step = [&](auto kc_c, auto rbc, auto sc) __attribute__((always_inline)) {
                constexpr int KC = decltype(kc_c)::value, rb = decltype(rbc)::value, s = decltype(sc)::value, n = rb * KS + s, n2 = n + PF;
                const int kd = KC >= 0 ? KC : kind;
                if constexpr (!(ABL & 4)) {
                    if constexpr (n2 < N) LV_READ(fb[n2 % NB], rbase[n2 % KS], (n2 / KS) * 16 * RB);
                    LV_WAIT(fb[n % NB], lv_min(PF, N - 1 - n));
                }
                __builtin_amdgcn_sched_barrier(0);
                lv_static_for<NT>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (ABL & 8) {
                        if constexpr (s == 0) accs[rb & 1][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    } else if constexpr (s == 0)
                        accs[rb & 1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[s][i], fb[n % NB], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else
                        accs[rb & 1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[s][i], fb[n % NB], accs[rb & 1][i], 0, 0, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (rb == MTB - 1 && !(ABL & 1)) {     // the next convolution's fragments out of LDS, as this one frees the registers
                    lv_static_for<NT>([&](auto ic) { LV_WREAD(W[s][ic], wrd, (s * FR + ic) * 1024) });
                    if constexpr (s == KS - 1 && NKS > KS)
                        if (kind == 2)
                            lv_static_for<NKS - KS>([&](auto s2c) { lv_static_for<NT>([&](auto ic) { LV_WREAD(W[KS + s2c][ic], wrd, ((KS + s2c) * FR + ic) * 1024) }); });
                }
                // (slot (rb + 1) % 3 held row block rb - 2, whose epilogue ran during row block rb - 1; this load is consumed during rb + 2)
                if constexpr (ACC && s == 0 && rb + 1 < MTB)
                    if (kd == 2) lv_static_for<NT>([&](auto ic) { pvb[(rb + 1) % 3][ic] = ld_prev(POST ? wso : tqo, rb + 1, ic * 32); });
                // (pieces start behind step 1: the first one reads accumulators that the previous row block's last MFMAs are still
                // writing when step 0 issues -- nine wait states)
                if constexpr (rb > 0 && !(ABL & 2))
                    lv_static_for<NP>([&](auto pc) {
                        if constexpr (1 + (decltype(pc)::value * (KS - 1)) / NP == s)
                            piece(kc_c, acc_c, last_c, std::integral_constant<int, rb - 1>{}, pc, kind, tqo, wso, wdst0, wdst1, xnext, tnext);
                    });
                __builtin_amdgcn_sched_barrier(0);
            };
            // Row blocks 0 .. MTB-2 with the kind a compile-time constant.  Whatever a path reads from LDS by inline asm it reads AND
            // waits for inside the path: at a join hipcc may copy a fragment register to where the other paths keep theirs, and to it
            // an asm load's destination is written when the statement ends -- a copy ahead of the wait moves bytes that have not
            // landed (seen: row block 0 of one wave in a few thousand garbage, only under load).  So the first reads are issued here,
            // behind the branch, and the path ends by draining the reads it has in flight for the last row block.
            auto main_part = [&](auto kc_c) __attribute__((always_inline)) {
                if constexpr (!(ABL & 4))
                    lv_static_for<PF>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        LV_READ(fb[n % NB], rbase[n % KS], (n / KS) * 16 * RB);
                    });
                __builtin_amdgcn_sched_barrier(0);
                lv_static_for<MTB - 1>([&](auto rbc) { lv_static_for<KS>([&](auto sc) { step(kc_c, rbc, sc); }); });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                lv_static_for<NB>([&](auto jc) { bf16x8_t d_ = fb[jc]; asm volatile("" : "+v"(d_)); fb[jc] = d_; });
                __builtin_amdgcn_sched_barrier(0);
            };
            if (kind == 0) main_part(std::integral_constant<int, 0>{});
            else if (kind == 1) main_part(std::integral_constant<int, 1>{});
            else main_part(std::integral_constant<int, 2>{});
            LV_STAMP(3)                                // [3] row blocks 0 .. MTB-2
            constexpr std::integral_constant<int, -1> any_kind{};
            lv_static_for<KS>([&](auto sc) { step(any_kind, std::integral_constant<int, MTB - 1>{}, sc); });
            LV_STAMP(4)                                // [4] last row block (+ fragment copies)
            if constexpr (!(ABL & 2))
                lv_static_for<NP>([&](auto pc) { piece(any_kind, acc_c, last_c, std::integral_constant<int, MTB - 1>{}, pc, kind, tqo, wso, wdst0, wdst1, xnext, tnext); });
            // the next convolution's fragments are in W before this body is left (the loop's back edge is a join too)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lv_static_for<KSMAX>([&](auto sc) { lv_static_for<NT>([&](auto ic) { bf16x8_t d_ = W[sc][ic]; asm volatile("" : "+v"(d_)); W[sc][ic] = d_; }); });
            __builtin_amdgcn_sched_barrier(0);
            cpar ^= 1;
            LV_STAMP(5)                                // [5] the last row block's epilogue (exposed)
            if constexpr ((ABL & 16) != 0) pf[6] += 1;                 // [6] convolutions
#undef LV_STAMP
        };

        // one residual block: xr holds the tile's raw input rows, W the fragments of its first convolution, the LDS buffer of the
        // current parity those of its second one
        auto block = [&](auto taps_c, auto nks_c, auto acc_c, auto last_c, const uint16_t *wb, const float *bb, const uint16_t *wnext_block,
                         int tile_nx) __attribute__((always_inline)) {
            constexpr int TAPS = decltype(taps_c)::value, KS = TAPS * KSUB, NKS = decltype(nks_c)::value;
            const unsigned long long tblk = (ABL & 16) ? __builtin_amdgcn_s_memtime() : 0;
            // x image = LeakyReLU(x), zero outside the sequence (the convolutions' zero padding).  Every wave is past the
            // convolution that read the previous x image (it met the others at the barrier of the one after).
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
                conv(taps_c, acc_c, nks_c, last_c, last ? 2 : (q & 1), dil ? X_OFF : M_OFF, dil ? GX : GM, dil ? q + 1 : 1, dil ? mw[0] : xw[0], dil ? mw[1] : xw[1],
                     bsb, q * C * 4, own2 ? wb : wnext_block, own2 ? (q + 2) * CB : (q - 4) * NKS * FR * 1024, own2 ? KS * FR : NKS * FR, xnext, tnext);
            }
        };

        constexpr int NBLK = (B0 > 0) + (B1 > 0) + (B2 > 0);
        // the block after the last one of this tile is the first one of the next tile
        block(std::integral_constant<int, B0>{}, std::integral_constant<int, (NBLK > 1 ? B1 : B0) * KSUB>{}, std::integral_constant<bool, ACC0>{}, std::integral_constant<bool, NBLK == 1>{}, p.w[0], p.bias[0],
              NBLK > 1 ? p.w[1] : p.w[0], NBLK > 1 ? tile : tile_next);
        if constexpr (B1 > 0)
            block(std::integral_constant<int, B1>{}, std::integral_constant<int, (NBLK > 2 ? B2 : B0) * KSUB>{}, std::true_type{}, std::integral_constant<bool, NBLK == 2>{}, p.w[1], p.bias[1],
                  NBLK > 2 ? p.w[2] : p.w[0], NBLK > 2 ? tile : tile_next);
        if constexpr (B2 > 0)
            block(std::integral_constant<int, B2>{}, std::integral_constant<int, B0 * KSUB>{}, std::true_type{}, std::true_type{}, p.w[2], p.bias[2], p.w[0], tile_next);
        if constexpr (POST) {
            // conv_post + tanh on the tile's mean, now in the x image (raw bf16 rows, stored rows +-3): the 7 x 32 products of an output
            // row added in k_conv_post's order (tap-major, channels ascending, one fma chain from the bias), taps beyond the sequence
            // skipped as there.  (The lanes of a wave read different rows of one slot: bank conflicts on every read -- a few per cent of a tile.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            uint16_t *arow = p.audio + (int64_t)b * T;
            const float pslope = p.post_slope;
            // one output row per thread and pass.  (Two adjacent rows per thread, so that a row's LeakyReLU is taken twice instead of
            // seven times, measured SLOWER: 1 511 against 1 449 us per 1 280 chunks -- six busy waves and rows 128 bytes apart.)
            for (int j = tid; j < R; j += NW * 64) {
                const int t = tq + HC + j;
                if (t >= T) break;
                float acc = p.post_bias;
#pragma unroll 1
                for (int k = 0; k < 7; k++) {
                    const int tt = t + k - 3;
                    if (tt < 0 || tt >= T) continue;
                    const int row = GX + HC + j + k - 3;
#pragma unroll
                    for (int q = 0; q < SPR; q++) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(lds + img_addr(X_OFF, row, q));
                        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            float lo = __uint_as_float(u[e] << 16), hi = __uint_as_float(u[e] & 0xffff0000u);
                            lo = lo > 0.0f ? lo : lo * pslope;
                            hi = hi > 0.0f ? hi : hi * pslope;
                            acc = __fmaf_rn(p.post_w[k * 32 + q * 8 + 2 * e], lo, acc);
                            acc = __fmaf_rn(p.post_w[k * 32 + q * 8 + 2 * e + 1], hi, acc);
                        }
                    }
                }
                arow[t] = f32_to_bf16(tanhf(acc));
            }
            __builtin_amdgcn_s_barrier();              // the next tile's first block refills the x image
        }
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

template <int C, int NW, int WGM, int WGN, int MTB, int HC, int PF, int B0, int B1, int B2, bool ACC0, int ABL = 0, bool POST = false>
static int launch_level(LevelParams &p, hipStream_t st, int64_t ws_bytes = 0)
{
    constexpr int RT = WGM * MTB * 16, R = RT - 2 * HC;
    constexpr int BMAX = B0 > B1 ? (B0 > B2 ? B0 : B2) : (B1 > B2 ? B1 : B2);
    constexpr size_t bytes = (size_t)(RT + 50 + RT + 10) * C * 2 + 2 * (size_t)BMAX * (C / 32) * (C / 16) * 1024;
    static_assert(bytes <= 160 * 1024, "tile does not fit in LDS");
    auto kern = k_resblock_level<C, NW, WGM, WGN, MTB, HC, PF, B0, B1, B2, ACC0, ABL, POST>;
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return check_hip(e, "resblock_level lds attr");
        attr_once.done(attr_dev);
    }
    p.tiles_per_seq = (p.T + R - 1) / R;
    p.ntiles = p.tiles_per_seq * p.nbatch;
    const int ncu = device_cu_count();
    if (ncu <= 0) return fail(IFH_EHIP, "resblock_level: device query");
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    if (POST && ws_bytes < (int64_t)grid * RT * C * 2) return fail(IFH_EINVAL, "resblock_level: mean_ws is smaller than ifh_level_ws_bytes()");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), bytes, st, p);
    return IFH_OK;
}

}  // namespace ifh

// bytes of ifh_level_desc.mean_ws: one tile of running-mean rows (896 x 32 bf16) per workgroup, a workgroup per CU
extern "C" int64_t ifh_level_ws_bytes(void)
{
    const int ncu = ifh::device_cu_count_physical();
    return ncu > 0 ? (int64_t)ncu * 896 * 32 * 2 : -1;
}

using namespace ifh;

extern "C" int ifh_resblock_level_bf16(const ifh_level_desc *d, ifh_stream_t stream)
{
    IFH_CHECK_ARG(d);
    IFH_CHECK_ARG(d->x && (d->out || d->post_w) && d->nblocks >= 1 && d->nblocks <= 3);
    IFH_CHECK_ARG(d->nbatch >= 0 && d->t >= 0);
    if (d->nbatch == 0 || d->t == 0) return IFH_OK;
    IFH_CHECK_ARG(d->c == 32);
    IFH_CHECK_ARG((((uintptr_t)d->x) & 7) == 0 && (((uintptr_t)d->out) & 7) == 0 && d->x_bstride % 4 == 0 && d->out_bstride % 4 == 0);
    IFH_CHECK_ARG(d->slope > 0.0f && d->slope <= 1.0f && (int64_t)d->nbatch * d->t < (1ll << 31));
    LevelParams p;
    p.x = (const uint16_t *)d->x;
    p.x_bstride = d->x_bstride;
    p.out = (uint16_t *)d->out;
    p.out_bstride = d->out_bstride;
    for (int j = 0; j < 3; j++) {
        p.w[j] = j < d->nblocks ? (const uint16_t *)d->wstream[j] : nullptr;
        p.bias[j] = j < d->nblocks ? d->bias[j] : nullptr;
        if (j < d->nblocks) IFH_CHECK_ARG(p.w[j] && p.bias[j] && (((uintptr_t)p.w[j]) & 15) == 0 && (((uintptr_t)p.bias[j]) & 15) == 0);
    }
    p.T = d->t;
    p.nbatch = d->nbatch;
    p.slope = d->slope;
    p.out_scale = d->out_scale;
    p.accumulate = d->accumulate;
    p.prof = (unsigned long long *)d->debug_prof;
    p.post_w = d->post_w;
    p.post_bias = d->post_bias;
    p.post_slope = d->post_slope;
    p.audio = (uint16_t *)d->audio;
    p.mean_ws = (uint16_t *)d->mean_ws;
    if (d->post_w) {
        // conv_post folded in: the three-block C = 32 level only, nothing to add to; `out` is not touched
        IFH_CHECK_ARG(d->audio && d->mean_ws && d->nblocks == 3 && !d->accumulate && d->post_slope > 0.0f && d->post_slope <= 1.0f);
        IFH_CHECK_ARG((((uintptr_t)d->mean_ws) & 15) == 0 && (((uintptr_t)d->post_w) & 3) == 0 && (((uintptr_t)d->audio) & 1) == 0);
    }
    hipStream_t st = as_stream(stream);
    int rc = IFH_EINVAL;
    const int t0 = d->taps[0], t1 = d->nblocks > 1 ? d->taps[1] : 0, t2 = d->nblocks > 2 ? d->taps[2] : 0;
    //                <C, NW, WGM, WGN, MTB, HC, PF, B0, B1, B2>
    // (tools builds may pick another wave count / prefetch depth: -DLV_NW=4 -DLV_WGM=4 -DLV_MTB=14 -DLV_PF=..; the tile stays 896 rows)
#ifndef LV_NW
#define LV_NW 8
#define LV_WGM 8
#define LV_MTB 7
#endif
#ifndef LV_PF
#define LV_PF 3
#endif
#define LV_SHAPE 32, LV_NW, LV_WGM, 1, LV_MTB, 64, LV_PF
#define LEVEL_CASE(A, B, C_)                                                                                    \
    if (t0 == A && t1 == B && t2 == C_)                                                                         \
        rc = d->accumulate ? launch_level<LV_SHAPE, A, B, C_, true>(p, st) : launch_level<LV_SHAPE, A, B, C_, false>(p, st);
#ifdef LV_WITH_PIPE        /* tools/mb/level_bench builds: the software-pipelined form of tools/exp/level_pipe.hip unless IFH_LEVEL_BARRIER is set */
    if (d->c == 32 && getenv("IFH_LEVEL_BARRIER") == nullptr) {
        rc = launch_level_pipe(p, t0, t1, t2, st);
        if (rc == IFH_EINVAL) return fail(IFH_EINVAL, "resblock_level: unsupported (c, taps) combination");
        if (rc != IFH_OK) return rc;
        IFH_LAUNCH_CHECK("resblock_level_bf16");
        return IFH_OK;
    }
#endif
    if (d->c == 32) {
#ifdef LV_DEV_ABL          /* tools builds: ablations of the k = 11 block, chosen by IFH_LEVEL_ABL (wrong results) */
        const int abl = getenv("IFH_LEVEL_ABL") ? atoi(getenv("IFH_LEVEL_ABL")) : 0;
#define LEVEL_ABL(V) if (abl == V && t0 == 11 && t1 == 0) return launch_level<LV_SHAPE, 11, 0, 0, false, V>(p, st);
        LEVEL_ABL(1) LEVEL_ABL(4) LEVEL_ABL(8) LEVEL_ABL(12) LEVEL_ABL(13) LEVEL_ABL(16)
        if (abl == 16 && t1 == 0 && t0 == 3) return launch_level<LV_SHAPE, 3, 0, 0, false, 16>(p, st);
        if (abl == 16 && t1 == 0 && t0 == 7) return launch_level<LV_SHAPE, 7, 0, 0, false, 16>(p, st);
#undef LEVEL_ABL
#endif
#ifdef LV_DEV_ONLY
        LEVEL_CASE(LV_DEV_ONLY, 0, 0)
#else
        if (d->post_w) {
            if (t0 == 3 && t1 == 7 && t2 == 11) rc = launch_level<LV_SHAPE, 3, 7, 11, false, 0, true>(p, st, d->mean_ws_bytes);
        } else
        LEVEL_CASE(3, 7, 11)
        LEVEL_CASE(3, 0, 0)
        LEVEL_CASE(7, 0, 0)
        LEVEL_CASE(11, 0, 0)
#endif
    }
#undef LEVEL_CASE
    if (rc == IFH_EINVAL) return fail(IFH_EINVAL, "resblock_level: unsupported (c, taps) combination");
    if (rc != IFH_OK) return rc;
    IFH_LAUNCH_CHECK("resblock_level_bf16");
    return IFH_OK;
}
