// seq.hip -- one whole HiFi-GAN residual block in a single launch over WHOLE sequences held in ONE LDS image:
//     for d in (1, 3, 5):  x = x + conv_k,1(lrelu(conv_k,d(lrelu(x))))          (k = taps, "same" padding)
//     out = x * out_scale  (+ out)
// (transformers modeling_speecht5.py HifiGanResidualBlock.forward, three per upsampling level, mean over the three in
// SpeechT5HifiGan.forward; reached from HelloSippyTTSRT/HelloSippyRTPipe.py:236).  Same arithmetic, k order and rounding points as
// k_resblock_chain (chain.hip) -- bit-identical to it -- with a different use of the LDS:
//
//   * k_resblock_chain keeps TWO operand images (x and the intermediate), so a tile is at most 384 rows at C = 64 and every tile
//     recomputes the rows its neighbours own (60 / 36 / 12 on either side for k = 11 / 7 / 3): 1.5 / 1.33 / 1.25 x the work
//     (measured round 5: the C = 64 chains run the matrix pipe as busy as the C = 128 ones, 50 % at k = 11, and take 1.5 x their time).
//   * Here a convolution's result OVERWRITES its operand: all MFMAs of a convolution read the image, the waves meet, every wave writes
//     the rows it owns (its accumulators are the whole result), the waves meet again.  One image holds a chunk's whole sequence
//     (768 rows at C = 64; two 192-row sequences at C = 128; two 48-row ones at C = 256 where the level's 18 convolutions used to be
//     18 launches with the activations crossing HBM in between): nothing is recomputed, only the weights stream.
//   * FOUR waves of up to 512 registers (one per SIMD) instead of eight of 256: a wave owns MT = 12 row tiles x NT = 4 channel tiles
//     (192 rows x 64 channels; C = 256: 3 x 8), i.e. 16 fragment reads per 48 MFMAs instead of 8 per 12 and four times the MFMAs
//     between two weight-unit barriers -- and it has the registers to keep the residual stream (accumulator layout, packed bf16) at
//     home: held in global memory instead, every epilogue moved 98 KB through the CU's ~10 B/clock global path (measured round 5: K
//     loops at 96 % of the matrix pipe's rate, 102 k of a tile's 227 k clocks in those loads and stores).
//   * The weights arrive as the pre-packed fragment stream of ops.w_chain_pack through a DMA ring -- as in chain.hip; the biases come
//     from global memory (L2) as an epilogue starts, which leaves the LDS to the image and a deeper ring.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "chain_util.h"

namespace ifh {

struct SeqParams {
    const uint16_t *x;
    int64_t x_bstride;
    const uint16_t *wstream;     // packed fragments of the 6 convolutions, padded to whole units
    const float *bias;           // [6][C]
    int nbatch, ntiles, nunits;
    float slope, out_scale;
    float post_slope;            // LeakyReLU on what is stored (1 = none): ifh_seq_desc.post_slope
    int accumulate;
    uint16_t *out;
    int64_t out_bstride;
    unsigned long long *prof;    // PROF builds (ifh_seq_desc.debug_prof): shader-clock sums of wave 0 of every workgroup per phase
};

template <int... I, class F>
__device__ __forceinline__ void seq_static_for_impl(std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void seq_static_for(F &&f)
{
    seq_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// Fragment reads.  A wave's MT row tiles are two halves of MH = MT / 2; a k-step runs as two HALF-STEPS (h = 0, 1), each NT passes
// (channel tiles) of MH MFMAs: MFMA k = i MH + j multiplies A_i (16 output channels x 32 k) with B_j of that half.  The activation
// fragments of a half are double-buffered per half-step: half-step (s, h) requests those of the next one -- (s, 1) or (s+1, 0) -- at
// slots seq_pb(b), all of them before its last pass.  The NT weight fragments of a k-step serve both halves and are refilled in place:
// A_(NT-1) of k-step s as (s, 0) starts (its register was in use until the end of (s-1, 1)), A_(i-1) of k-step s+1 as pass i >= 1 of
// (s, 1) starts.  Every fragment is requested >= (NT-1) MH MFMAs before its first use.  LDS reads return in order, so the wait in front
// of an MFMA is "no more outstanding than the reads requested after the fragment it needs": counted here at compile time.
constexpr int seq_pb(int b, int mh, int nt) { return 1 + b * ((nt - 1) * mh - 1) / mh; }
constexpr int seq_b_at(int k, int mh, int nt)
{
    for (int b = 0; b < mh; b++)
        if (seq_pb(b, mh, nt) == k) return b;
    return -1;
}
// does a step of type h request a weight fragment at slot k (in front of MFMA k)?  Type 2 = a FULL step (one step per k-step over all
// of a wave's row tiles, mh = MT: it is its own predecessor and makes the requests of both half-step types).
constexpr bool seq_a_at(int h, int k, int mh) { return k % mh == 0 && (h == 0 ? k == 0 : h == 1 ? k > 0 : true); }
// number of reads a half-step of type h requests at slots < k (bnext: it requests the next half-step's activation fragments;
// anext: a type-1 half-step requests the next k-step's weight fragments)
constexpr int seq_issued(int h, int k, int mh, int nt, bool bnext, bool anext)
{
    int n = 0;
    for (int q = 0; q < k && q < nt * mh; q++) {
        if (seq_a_at(h, q, mh) && (h == 0 || (h == 2 && q == 0) || anext)) n++;
        if (bnext && seq_b_at(q, mh, nt) >= 0) n++;
    }
    return n;
}
// ordinal of the weight (is_a) / activation read requested at `slot` by a (full) half-step of type h
constexpr int seq_ord(int h, bool is_a, int slot, int mh, int nt)
{
    int n = 0;
    for (int q = 0; q < nt * mh; q++) {
        if (seq_a_at(h, q, mh)) {
            if (is_a && q == slot) return n;
            n++;
        }
        if (seq_b_at(q, mh, nt) >= 0) {
            if (!is_a && q == slot) return n;
            n++;
        }
    }
    return n;
}
// outstanding reads allowed in front of MFMA k = i MH + j of a half-step of type h (its predecessor, of type 1 - h, requested
// everything a full half-step requests: the prologue of a convolution does the same)
constexpr int seq_allow(int h, int k, int mh, int nt, bool bnext, bool anext)
{
    const int i = k / mh, j = k % mh, done = seq_issued(h, k, mh, nt, bnext, anext);
    const int hp = h == 2 ? 2 : 1 - h;                                                                  // the predecessor's type
    const int nprev = seq_issued(hp, nt * mh, mh, nt, true, true);
    int allow = (nprev - 1 - seq_ord(hp, false, seq_pb(j, mh, nt), mh, nt)) + done;                    // B_j
    if (h != 1) {
        const int a = (i == nt - 1) ? done - 1 : (nprev - 1 - seq_ord(h == 2 ? 2 : 1, true, (i + 1) * mh, mh, nt)) + done;
        allow = a < allow ? a : allow;
    }
    return allow;
}

// C channels; TAPS; TSEQ rows per sequence, NSEQ sequences per tile; 4 waves = WGM row groups x WGN channel groups, a wave owns MT row
// tiles x NT channel tiles of 16 x 16 as two halves of MT / 2 row tiles: consecutive rows of ONE sequence, or (HSEQ) the same rows of
// the tile's TWO sequences; NRING weight units of UNITB bytes in the ring.
template <int C, int TAPS, int TSEQ, int NSEQ, int NW, int WGN, int NT, int MT, int NH, bool HSEQ, int NRING, int UNITB, bool ACCUM, bool PROF>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void k_resblock_seq(const SeqParams p)
{
    // diagnostic phase clocks (PROF instantiation, chosen when ifh_seq_desc.debug_prof is set): wave 0 sums the shader clocks spent in
    // [0] tile top (x rows -> image), [1] K loops, [2] the wait for the other waves behind a K loop, [3] conv1 epilogues, [4] conv2
    // epilogues, [5] the last epilogue; [6] tiles, [7] workgroup lifetime -- added to p.prof once, as the workgroup ends
    unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = PROF ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long tstart = tprev;
#define SEQ_STAMP(IDX)                                                     \
    if constexpr (PROF) {                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        pf[IDX] += now_ - tprev;                                           \
        tprev = now_;                                                      \
    }
    static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
    constexpr int WGM = NW / WGN;
    static_assert(NH == 1 || NH == 2, "a k-step over all of a wave's row tiles at once, or as two half-steps");
    constexpr int MH = MT / NH;                        // row tiles per step
    constexpr int MS = HSEQ ? MT / 2 : MT;             // row tiles of ONE sequence a wave owns
    constexpr int RW = MS * 16;                        // rows of a sequence a wave owns
    constexpr int WPS = TSEQ / RW;                     // row groups per sequence
    static_assert(MT % NH == 0 && MT % 2 == 0 && WGM * WGN == NW && WGN * NT * 16 == C && TSEQ % RW == 0, "tile shape");
    static_assert(HSEQ ? (NSEQ == 2 && WGM == WPS) : (WGM == NSEQ * WPS), "row groups");
    static_assert(NRING >= 2 && UNITB % (NW * 1024) == 0, "ring");
    constexpr int SB = (C + 16) * 2;                   // row stride: C*2 + 32 bytes = 2 (mod 4) sixteen-byte slots -> conflict-free fragment reads
    constexpr int H = (TAPS - 1) / 2;
    constexpr int GX = 5 * H;                          // reach of the dilation-5 convolution
    constexpr int SROWS = TSEQ + GX;                   // sequence pitch: the guard rows between two sequences are both sequences' zero padding
    constexpr int XROWS = NSEQ * SROWS + GX;
    constexpr int KSUB = C / 32, KS = TAPS * KSUB, FRAGS = C / 16;
    constexpr int UK = UNITB / (FRAGS * 1024);         // k-steps per unit
    static_assert(UK >= 1 && UK * FRAGS * 1024 == UNITB, "a unit is a whole number of k-steps");
    constexpr int PIECES = UNITB / (NW * 1024);        // DMA instructions per wave and unit
    constexpr int RING_OFF = XROWS * SB;
    constexpr int BIAS_OFF = RING_OFF + NRING * UNITB;
    // the biases live in LDS where it has room for them (C <= 128); else (C = 256) they come from global memory as a K loop ends.  (Not
    // everywhere: a global load is one more vector-memory operation in front of the ring's counted vmcnt waits, which then wait for
    // the youngest weight DMA too -- at 6 k-steps per convolution that stall was the larger part of a K loop.)
    constexpr bool BIAS_LDS = BIAS_OFF + 6 * C * 4 <= 160 * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int wm = wid % WGM, wn = wid / WGM;
    const int sq0 = HSEQ ? 0 : wm / WPS;               // the sequence of this wave's first half (HSEQ: the second half is sequence 1)
    const int srow = (HSEQ ? wm : wm % WPS) * RW + fr; // its lane's first row inside the sequence
    // row tile j of the wave (j >= MH: second half): byte offset in the image from row tile 0; its sequence; its row tile inside that
    struct J {
        static constexpr int off(int j) { return HSEQ ? (j / MS) * SROWS * SB + (j % MS) * 16 * SB : j * 16 * SB; }
        static constexpr int seq(int j) { return HSEQ ? j / MS : 0; }
        static constexpr int row(int j) { return HSEQ ? j % MS : j; }
    };
    constexpr int NSD = HSEQ ? 2 : 1;                  // sequences (buffer descriptors) per wave

    for (int i = tid * 16; i < RING_OFF; i += NW * 64 * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0, 0, 0, 0);
    if constexpr (BIAS_LDS)
        for (int i = tid; i < 6 * C; i += NW * 64) reinterpret_cast<float *>(lds + BIAS_OFF)[i] = p.bias[i];
    const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(p.wstream) + wid * (UNITB / NW) + lane * 16;
    int u_issue = 0, ring_issue = 0, ring_read = 0, ks_in_unit = 0;
#define SEQ_DMA()                                                                                                          \
    {                                                                                                                      \
        _Pragma("unroll") for (int h_ = 0; h_ < PIECES; h_++)                                                              \
            __builtin_amdgcn_global_load_lds(                                                                              \
                (const __attribute__((address_space(1))) void *)(wsrc + (int64_t)u_issue * UNITB + h_ * 1024),             \
                (__attribute__((address_space(3))) void *)(lds + RING_OFF + ring_issue * UNITB + wid * (UNITB / NW) + h_ * 1024), 16, 0, 0); \
        u_issue = (u_issue + 1 == p.nunits) ? 0 : u_issue + 1;                                                             \
        ring_issue = (ring_issue + 1 == NRING) ? 0 : ring_issue + 1;                                                       \
    }
    constexpr int AHEAD = NH == 2 ? NRING - 1 : NRING - 2;       // units in flight beyond the one being entered (see the protocol below)
    static_assert(AHEAD >= 1, "ring depth");
#pragma unroll
    for (int d = 0; d < AHEAD; d++) SEQ_DMA()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // the image is zero before any wave writes its rows

    // Weight-ring protocol.  Half-steps (NH = 2): the ring holds units v .. v+NRING-1; ENTERING unit v+1 happens as half-step 1 of the
    // last k-step of unit v starts (its passes request unit v+1's first fragments): a wave waits for its own pieces of unit v+1 (all DMAs
    // but the NRING-2 youngest units'), meets the others -- unit v+1 is then complete for everybody and nobody reads unit v again: every
    // weight fragment of the running k-step is in registers -- and refills the slot of unit v with unit v+NRING.  Full steps (NH = 1):
    // the entering step still takes its own last weight fragment from unit v, so the slot refilled is the one of unit v-1 and one unit
    // less is in flight.  No LDS drain in front of the barrier: the reads of the slot that is refilled were waited for, one by one, in
    // front of the MFMAs that used them.
#define SEQ_ENTER_WAIT()                                                                \
    {                                                                                   \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PIECES) : "memory");     \
        __builtin_amdgcn_s_barrier();                                                   \
    }

    const int xb = (GX + sq0 * SROWS + srow) * SB + fg * 16;             // B-fragment base of row tile 0 (tap offset and k sub-step added per k-step)
    const int ab = RING_OFF + wn * NT * 1024 + lane * 16;                // A-fragment base inside a k-step of the ring
    const int cw = (wn * NT * 16 + 4 * fg) * 2;                          // byte offset of this lane's 4 channels in a row (tile 0)
    const int xw = (GX + sq0 * SROWS + srow) * SB + cw;                  // store base of this lane
    const float slope = p.slope, out_scale = p.out_scale, post_slope = p.post_slope;

    uint2 xr[MT][NT];                                  // residual stream of the owned elements, packed bf16 (accumulator layout)
    f32x4 acc[NT][MT];
    // Global rows go through buffer instructions: a wave-uniform descriptor per sequence, ONE per-lane byte offset (row srow, this lane's
    // channels) and a scalar offset per (row tile, channel tile).  (As 64-bit pointers hipcc keeps an address pair per row tile alive
    // across the six convolutions: level.hip.)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    struct Srds {
        __amdgpu_buffer_rsrc_t r[NSD];
    };
    // (a short last tile recomputes the last sequence and stores nothing: `guard` gives a missing sequence a descriptor of zero
    // records -- the hardware drops out-of-range buffer stores -- so the epilogue needs no branch)
    auto srds = [&](const uint16_t *base, int64_t bstride, int tl, bool guard) __attribute__((always_inline)) {
        Srds o;
#pragma unroll
        for (int h = 0; h < NSD; h++) {
            const int b = tl * NSEQ + sq0 + h;
            o.r[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(base + (int64_t)min(b, p.nbatch - 1) * bstride), 0,
                                                       (guard && b >= p.nbatch) ? 0 : 0x7ffffffc, 0x00020000);
        }
        return o;
    };
    int voff = srow * C * 2 + cw;
    auto ld_row = [&](const Srds &r, int j, int i) __attribute__((always_inline)) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r.r[J::seq(j)], voff, J::row(j) * 16 * C * 2 + i * 32, 0);
        return make_uint2(v.x, v.y);
    };
    auto st_row = [&](const Srds &r, int j, int i, uint2 v) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b64((u32x2){v.x, v.y}, r.r[J::seq(j)], voff, J::row(j) * 16 * C * 2 + i * 32, 0);
    };
    // The global loads of an epilogue must not be hoisted into the K loop in front of it (more live registers there made hipcc spill
    // fragment registers right behind their asm reads, i.e. before the data had arrived): their address is re-made opaque here.
#define SEQ_PIN_GLOBAL() asm volatile("" : "+v"(voff)::"memory");
    const __amdgpu_buffer_rsrc_t brd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bias), 0, 6 * C * 4, 0x00020000);
    const int bvoff = (wn * NT * 16 + 4 * fg) * 4;
    // this lane's biases of convolution q (L2; requested as the K loop's last MFMAs run, used behind the barrier)
    auto read_bias = [&](int q, f32x4 (&bv)[NT]) __attribute__((always_inline)) {
        if constexpr (!BIAS_LDS) {
#pragma unroll
            for (int i = 0; i < NT; i++) bv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brd, bvoff, q * C * 4 + i * 64, 0));
        }
    };
    // (LDS form: one channel tile's four biases at a time, as an epilogue reaches that tile -- the epilogues walk channel tile outer, row
    // tile inner, so 4 bias registers are live instead of 16.  Read AND wait in ONE asm statement: to hipcc an asm load's destination is
    // written when the statement ends, and under register pressure it spilled a bias register to scratch between a read and a separate
    // wait, i.e. before the data had arrived -- tools/lint_asm_loads.py looks for exactly that in the listing.)
    auto bias_tile = [&](int q, int i, const f32x4 (&bv)[NT]) __attribute__((always_inline)) {
        if constexpr (BIAS_LDS) {
            // (the lane's base through an opaque copy: hipcc otherwise keeps base + constant for every (convolution, tile) it can see
            // in a register of its own across the K loops -- nine of them -- and spills others for it: 45-66 spills per kernel, whose
            // reloads at every pass start sat behind an s_waitcnt vmcnt(0), i.e. drained the weight ring)
            int bb_ = bvoff;
            asm volatile("" : "+v"(bb_));
            const int addr = BIAS_OFF + bb_ + q * C * 4 + i * 64;
            f32x4 b;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b) : "v"(addr) : "memory");
            return b;
        } else {
            return bv[i];
        }
    };
    int tile = blockIdx.x;
    if (tile < p.ntiles) {
        const Srds xs = srds(p.x, p.x_bstride, tile, false);
#pragma unroll
        for (int j = 0; j < MT; j++)
#pragma unroll
            for (int i = 0; i < NT; i++) xr[j][i] = ld_row(xs, j, i);
    }

    int cur_a = 0, cur_b = 0, nxt_a = 0, nxt_b = 0;
    bool nxt_enter = false;
    // addresses of k-step s of the current convolution (no synchronisation: it can sit behind MFMAs)
    auto advance = [&](int dd, int s) {
        nxt_enter = ks_in_unit == 0;
        const int tap = s / KSUB, cs = s - tap * KSUB;
        int ab_ = ab, xb_ = xb;                            // (opaque copies: see bias_tile)
        asm volatile("" : "+v"(ab_), "+v"(xb_));
        nxt_a = ab_ + ring_read * UNITB + ks_in_unit * (FRAGS * 1024);
        nxt_b = xb_ + (tap - H) * dd * SB + cs * 64;
        if (++ks_in_unit == UK) {
            ks_in_unit = 0;
            ring_read = (ring_read + 1 == NRING) ? 0 : ring_read + 1;
        }
    };
// (clang: an asm operand inside a nested generic lambda cannot name a captured variable -- hence the local copies)
#define SEQ_READ(DST, ADDR, OFF)                                                                                     \
    {                                                                                                                \
        const int a_ = (ADDR);                                                                                       \
        bf16x8_t d_;                                                                                                 \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d_) : "v"(a_), "n"(OFF));                                \
        DST = d_;                                                                                                    \
    }
    // One half-step: NT passes of MH MFMAs on (A_i, B_0..B_(MH-1) of half HALF); fa[i] holds A_i, cb the current activation fragments, nb
    // receives the next half-step's.  If the next k-step opens a weight unit the waves meet as half-step 1 starts -- nobody reads the
    // running k-step's unit any more, so the refill DMA (behind the second MFMA: between a barrier and the first MFMA the matrix pipe
    // idles) may take the oldest slot.
    bf16x8_t fa[NT];
    auto hstep = [&](const bf16x8_t (&cb)[MH], bf16x8_t (&nb)[MH], auto half_c, auto has_next, auto is_first) {
        // NEXT: a k-step follows this one; FIRST: the accumulators start from zero
        constexpr int HALF = decltype(half_c)::value;                    // 0 / 1: first / second half-step; 2: a full step
        constexpr bool NEXT = decltype(has_next)::value, FIRST = decltype(is_first)::value;
        constexpr bool BNEXT = HALF == 0 || NEXT;                        // this step requests a successor's activation fragments
        constexpr int RB = HALF == 1 ? MH : 0;                           // its first row tile
        const bool enter = HALF != 0 && NEXT && nxt_enter;
        if (enter) SEQ_ENTER_WAIT()
        __builtin_amdgcn_sched_barrier(0);
        // (every schedule quantity below is a compile-time constant of k: as a run-time loop variable hipcc evaluated the constexpr
        // schedule functions on the scalar unit, per MFMA)
        seq_static_for<NT * MH>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value, i = k / MH, j = k - i * MH;
            constexpr int allow = seq_allow(HALF, k, MH, NT, BNEXT, NEXT), bn = seq_b_at(k, MH, NT);
            if constexpr (i == 0 || (j == 0 && HALF != 1)) wait_lgkm(allow);                            // A_i / B_j are back
            if constexpr (j == 0) {
                if constexpr (HALF != 1 && i == 0) SEQ_READ(fa[NT - 1], cur_a, (NT - 1) * 1024)
                else if constexpr (HALF != 0 && i > 0 && NEXT) SEQ_READ(fa[i - 1], nxt_a, (i - 1) * 1024)
            }
            if constexpr (BNEXT && bn >= 0) {
                if constexpr (HALF == 0) SEQ_READ(nb[bn], cur_b, J::off(MH + bn))
                else SEQ_READ(nb[bn], nxt_b, J::off(bn))
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FIRST)
                acc[i][RB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], cb[j], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else
                acc[i][RB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], cb[j], acc[i][RB + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (k == 1) {
                if (enter) {
                    SEQ_DMA()
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        });
        // The accumulators are "used" here: outside the k loop nothing else needs an MFMA's result before the epilogue, and hipcc sank the
        // MFMAs of the last steps below the next step's asm reads -- their fragment registers then had to outlive the reads that refill
        // them, and the extra pressure was spilled right behind asm reads (tools/lint_asm_loads.py).  (No instruction is emitted.)
        seq_static_for<NT>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            static_assert(MH == 3 || MH == 6 || MH == 12, "row tiles per step");
            seq_static_for<MH / 3>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = RB + 3 * decltype(gc)::value;
                const f32x4 u0 = acc[i][g], u1 = acc[i][g + 1], u2 = acc[i][g + 2];      // (an asm operand here cannot name a captured variable)
                if constexpr (NW == 4) asm volatile("" ::"a"(u0), "a"(u1), "a"(u2));      // (where they live: see SEQ_ACC_HERE)
                else asm volatile("" ::"v"(u0), "v"(u1), "v"(u2));
            });
        });
    };
    // one convolution over the image with dilation dd; the biases of convolution q are requested as its last half-step starts
    auto conv = [&](int dd, int q, f32x4 (&bv)[NT]) {
        bf16x8_t fb0[MH], fb1[MH];
        advance(dd, 0);
        if (nxt_enter) {
            SEQ_ENTER_WAIT()
            SEQ_DMA()
        }
        // the first half-step's fragments, requested in the order a type-1 half-step requests its successor's (same counted waits)
        seq_static_for<NT * MH>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value, bn = seq_b_at(k, MH, NT);
            if constexpr (k % MH == 0 && k > 0) SEQ_READ(fa[k / MH - 1], nxt_a, (k / MH - 1) * 1024)
            if constexpr (bn >= 0) SEQ_READ(fb0[bn], nxt_b, J::off(bn))
        });
        __builtin_amdgcn_sched_barrier(0);
        static_assert(KS >= 4 && KS % 2 == 0, "first, middle and last k-steps; full steps alternate two fragment sets");
        constexpr std::integral_constant<int, 0> h0{};
        constexpr std::integral_constant<int, 1> h1{};
        constexpr std::integral_constant<int, 2> hf{};
        constexpr std::true_type yes{};
        constexpr std::false_type no{};
        if constexpr (NH == 2) {
            // k-step s: (s, 0) on (cur_a, cur_b), then the addresses of k-step s + 1, then (s, 1)
            cur_a = nxt_a;
            cur_b = nxt_b;
            hstep(fb0, fb1, h0, yes, yes);
            advance(dd, 1);
            hstep(fb1, fb0, h1, yes, yes);
#pragma unroll 1
            for (int s = 1; s + 1 < KS; s++) {
                cur_a = nxt_a;
                cur_b = nxt_b;
                hstep(fb0, fb1, h0, yes, no);
                advance(dd, s + 1);
                hstep(fb1, fb0, h1, yes, no);
            }
            cur_a = nxt_a;
            cur_b = nxt_b;
            hstep(fb0, fb1, h0, no, no);
            read_bias(q, bv);
            hstep(fb1, fb0, h1, no, no);
        } else {
            // full steps: the addresses of k-step s + 1 are set before step s starts (it requests that step's fragments)
            cur_a = nxt_a;
            advance(dd, 1);
            hstep(fb0, fb1, hf, yes, yes);
            cur_a = nxt_a;
            advance(dd, 2);
            hstep(fb1, fb0, hf, yes, no);
#pragma unroll 1
            for (int s = 2; s + 2 < KS; s += 2) {
                cur_a = nxt_a;
                advance(dd, s + 1);
                hstep(fb0, fb1, hf, yes, no);
                cur_a = nxt_a;
                advance(dd, s + 2);
                hstep(fb1, fb0, hf, yes, no);
            }
            cur_a = nxt_a;
            advance(dd, KS - 1);
            hstep(fb0, fb1, hf, yes, no);
            cur_a = nxt_a;
            read_bias(q, bv);
            hstep(fb1, fb0, hf, no, no);
        }
        // every wave is done reading the image (its last fragments are in registers): the result may overwrite it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEQ_STAMP(1)
        __builtin_amdgcn_s_barrier();
        SEQ_STAMP(2)
    };
// An accumulator tile stays in the accumulator registers until the epilogue takes it: without this hipcc copied all 48 tiles to vector
// registers behind the last MFMA -- 192 registers beside the 96 of the residual stream -- and spilled the residual stream to scratch.
// (Four waves only: with two waves per SIMD an "a" operand makes hipcc split the 256 registers 128 / 128.)
#define SEQ_ACC_HERE(ACC)                                 \
    if constexpr (NW == 4) {                              \
        f32x4 t_ = (ACC);                                 \
        asm volatile("" : "+a"(t_));                      \
        (ACC) = t_;                                       \
    }
#define SEQ_LDS_STORE(ADDR, OFF, VAL)                                                                                \
    {                                                                                                                \
        const uint2 v_ = (VAL);                                                                                      \
        const int a_ = (ADDR);                                                                                       \
        const unsigned long long q_ = ((unsigned long long)v_.y << 32) | v_.x;                                       \
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a_), "v"(q_), "n"(OFF) : "memory");                       \
    }
    // conv1 epilogue: + bias, round to bf16 (what a separate launch stores), LeakyReLU (what the next one applies on load) -> the image
    auto epi1 = [&](int q, const f32x4 (&bv)[NT]) {
        f32x4 bt;
        seq_static_for<MT * NT>([&](auto tc) __attribute__((always_inline)) {
            constexpr int i = decltype(tc)::value / MT, j = decltype(tc)::value % MT;
            if constexpr (j == 0) bt = bias_tile(q, i, bv);
            SEQ_ACC_HERE(acc[i][j])
            const f32x4 a = acc[i][j];
            uint2 pk = make_uint2(f32x2_to_bf16x2(a[0] + bt[0], a[1] + bt[1]), f32x2_to_bf16x2(a[2] + bt[2], a[3] + bt[3]));
            pk = chain_lrelu4(pk, slope);
            SEQ_LDS_STORE(xw, J::off(j) + i * 32, pk)
            __builtin_amdgcn_sched_barrier(0);         // tile by tile: interleaved for ILP the tiles' temporaries pushed other values to scratch
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SEQ_STAMP(3)
    };
    // conv2 epilogue: + bias + residual -> new residual (registers) and its LeakyReLU'd image
    auto epi2 = [&](int q, const f32x4 (&bv)[NT]) {
        f32x4 bt;
        seq_static_for<MT * NT>([&](auto tc) __attribute__((always_inline)) {
            constexpr int i = decltype(tc)::value / MT, j = decltype(tc)::value % MT;
            if constexpr (j == 0) bt = bias_tile(q, i, bv);
            SEQ_ACC_HERE(acc[i][j])
            const f32x4 a = acc[i][j];
            const uint2 rv = xr[j][i];
            float v0 = a[0] + bt[0], v1 = a[1] + bt[1], v2 = a[2] + bt[2], v3 = a[3] + bt[3];
            v0 += __uint_as_float(rv.x << 16);
            v1 += __uint_as_float(rv.x & 0xffff0000u);
            v2 += __uint_as_float(rv.y << 16);
            v3 += __uint_as_float(rv.y & 0xffff0000u);
            const uint2 pk = make_uint2(f32x2_to_bf16x2(v0, v1), f32x2_to_bf16x2(v2, v3));
            xr[j][i] = pk;
            SEQ_LDS_STORE(xw, J::off(j) + i * 32, chain_lrelu4(pk, slope))
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SEQ_STAMP(4)
    };

    for (; tile < p.ntiles; tile += gridDim.x) {
        // the image = LeakyReLU(x); the guard rows are zero for good (the convolutions' zero padding)
        seq_static_for<MT * NT>([&](auto tc) __attribute__((always_inline)) {
            constexpr int j = decltype(tc)::value / NT, i = decltype(tc)::value % NT;
            SEQ_LDS_STORE(xw, J::off(j) + i * 32, chain_lrelu4(xr[j][i], slope))
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SEQ_STAMP(0)
        f32x4 bv[NT];
#pragma unroll 1
        for (int pr = 0; pr < 2; pr++) {
            conv(2 * pr + 1, 2 * pr, bv);
            epi1(2 * pr, bv);
            conv(1, 2 * pr + 1, bv);
            epi2(2 * pr + 1, bv);
        }
        conv(5, 4, bv);
        epi1(4, bv);
        conv(1, 5, bv);
        // last epilogue: (+ bias + residual) * out_scale (+ out) -> global, free of branches (around each load a branch made hipcc
        // spill the loaded value behind a full vmcnt(0): 130 serialised round trips per tile).  The rows of `out` it adds to are requested
        // PD tiles ahead; the residual registers take the next tile's rows (the last tile: its own again) as they fall free.  (All 24
        // tiles' rows of `out` requested at once and added in a second pass measured slower: 54 k clocks against 40 k.)
        {
            SEQ_PIN_GLOBAL()
            const int tnext = tile + (int)gridDim.x;
            const Srds os = srds(p.out, p.out_bstride, tile, true);
            const Srds xn = srds(p.x, p.x_bstride, tnext < p.ntiles ? tnext : tile, false);
            constexpr int PD = 8;                                        // tiles of `out` requested ahead (16 registers)
            static_assert(PD <= MT * NT, "prefetch distance");
            uint2 pv[PD];
            // Row tile outer, channel tile inner (round 6; the other epilogues walk channel tile outer): the NT pieces of a row -- 32 bytes
            // each, one store instruction each -- then leave back to back and meet in L2 as whole 64-byte sectors.  Channel tile outer they
            // left MT tiles apart, the L2s wrote the half-filled lines back in between, and the launch wrote 1.75-2.1 x its output
            // (TCC_EA0_WRREQ: 28 % of the write requests 32-byte ones; k_resblock_chain, whose pieces leave together: none).
            auto tj = [](int t) constexpr { return t / NT; };
            auto ti = [](int t) constexpr { return t % NT; };
            if constexpr (ACCUM) {
#pragma unroll
                for (int t = 0; t < PD; t++) pv[t] = ld_row(os, tj(t), ti(t));
            }
            f32x4 bts[NT];
#pragma unroll
            for (int i = 0; i < NT; i++) bts[i] = bias_tile(5, i, bv);
            seq_static_for<MT * NT>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value, i = t % NT, j = t / NT;
                const f32x4 bt = bts[i];
                SEQ_ACC_HERE(acc[i][j])
                f32x4 a = acc[i][j];
                const uint2 rv = xr[j][i];
                a[0] = (a[0] + bt[0] + __uint_as_float(rv.x << 16)) * out_scale;
                a[1] = (a[1] + bt[1] + __uint_as_float(rv.x & 0xffff0000u)) * out_scale;
                a[2] = (a[2] + bt[2] + __uint_as_float(rv.y << 16)) * out_scale;
                a[3] = (a[3] + bt[3] + __uint_as_float(rv.y & 0xffff0000u)) * out_scale;
                xr[j][i] = ld_row(xn, j, i);
                if constexpr (ACCUM) {
                    const uint2 q2 = pv[t % PD];
                    a[0] += __uint_as_float(q2.x << 16);
                    a[1] += __uint_as_float(q2.x & 0xffff0000u);
                    a[2] += __uint_as_float(q2.y << 16);
                    a[3] += __uint_as_float(q2.y & 0xffff0000u);
                    if constexpr (t + PD < MT * NT) pv[t % PD] = ld_row(os, (t + PD) / NT, (t + PD) % NT);
                }
                uint2 pk = make_uint2(f32x2_to_bf16x2(a[0], a[1]), f32x2_to_bf16x2(a[2], a[3]));
                if (post_slope != 1.0f) {              // the consumer's LeakyReLU-on-load, taken here (same bits); a wave-uniform branch
                    asm volatile("" ::: "memory");
                    pk = chain_lrelu4(pk, post_slope);
                }
                st_row(os, j, i, pk);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if constexpr (PROF) {
            SEQ_STAMP(5)
            pf[6] += 1;
        }
        // the weight stream restarts with every tile: what is left of a partly read unit is padding
        if (ks_in_unit != 0) {
            ks_in_unit = 0;
            ring_read = (ring_read + 1 == NRING) ? 0 : ring_read + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (PROF) {
        pf[7] = __builtin_amdgcn_s_memtime() - tstart;
        if (tid == 0)
            for (int i = 0; i < 8; i++) atomicAdd(p.prof + i, pf[i]);
    }
#undef SEQ_STAMP
#undef SEQ_READ
#undef SEQ_LDS_STORE
#undef SEQ_PIN_GLOBAL
#undef SEQ_ACC_HERE
#undef SEQ_ENTER_WAIT
#undef SEQ_DMA
}

template <int C, int TAPS, int TSEQ, int NSEQ, int NW, int WGN, int NT, int MT, int NH, bool HSEQ, int NRING, int UNITB>
static int launch_seq(SeqParams &p, hipStream_t st)
{
    constexpr int SB = (C + 16) * 2, GX = 5 * (TAPS - 1) / 2, XROWS = NSEQ * (TSEQ + GX) + GX;
    constexpr int FRAGS = C / 16, UK = UNITB / (FRAGS * 1024);
    constexpr size_t bytes0 = (size_t)XROWS * SB + (size_t)NRING * UNITB;
    constexpr size_t bytes = bytes0 + 6 * C * 4 <= 160 * 1024 ? bytes0 + 6 * C * 4 : bytes0;         // (the biases where they fit)
    static_assert(bytes <= 160 * 1024, "tile does not fit in LDS");
    const int ksteps = 6 * TAPS * (C / 32);
    if (p.nunits != (ksteps + UK - 1) / UK) return fail(IFH_EINVAL, "resblock_seq: weight stream length does not match c/taps/unit");
    auto kern0 = k_resblock_seq<C, TAPS, TSEQ, NSEQ, NW, WGN, NT, MT, NH, HSEQ, NRING, UNITB, false, false>;
    auto kern1 = k_resblock_seq<C, TAPS, TSEQ, NSEQ, NW, WGN, NT, MT, NH, HSEQ, NRING, UNITB, true, false>;
    auto kern_prof = k_resblock_seq<C, TAPS, TSEQ, NSEQ, NW, WGN, NT, MT, NH, HSEQ, NRING, UNITB, true, true>;      // (phase clocks: the accumulating form only)
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern0, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)kern1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)kern_prof, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return check_hip(e, "resblock_seq lds attr");
        attr_once.done(attr_dev);
    }
    p.ntiles = (p.nbatch + NSEQ - 1) / NSEQ;
    const int ncu = device_cu_count();
    if (ncu <= 0) return fail(IFH_EHIP, "resblock_seq: device query");
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    if (p.prof && p.accumulate)
        hipLaunchKernelGGL(kern_prof, dim3(grid), dim3(NW * 64), bytes, st, p);
    else if (p.accumulate)
        hipLaunchKernelGGL(kern1, dim3(grid), dim3(NW * 64), bytes, st, p);
    else
        hipLaunchKernelGGL(kern0, dim3(grid), dim3(NW * 64), bytes, st, p);
    return IFH_OK;
}

}  // namespace ifh

using namespace ifh;

extern "C" int ifh_resblock_seq_unit_bytes(int c) { return c == 256 ? 16384 : 8192; }

extern "C" int ifh_resblock_seq_supported(int c, int t, int taps)
{
    const bool shape = (c == 64 && t == 768) || (c == 128 && t == 192) || (c == 256 && t == 48);
    return shape && (taps == 7 || taps == 11 || (taps == 3 && c != 64)) ? 1 : 0;
}

extern "C" int ifh_resblock_seq_bf16(const ifh_seq_desc *d, ifh_stream_t stream)
{
    IFH_CHECK_ARG(d);
    IFH_CHECK_ARG(d->x && d->wstream && d->bias && d->out);
    IFH_CHECK_ARG(d->nbatch >= 0 && d->t >= 0);
    if (d->nbatch == 0 || d->t == 0) return IFH_OK;
    IFH_CHECK_ARG((d->c == 64 && d->t == 768) || (d->c == 128 && d->t == 192) || (d->c == 256 && d->t == 48));
    IFH_CHECK_ARG(d->taps == 7 || d->taps == 11 || (d->taps == 3 && d->c != 64));
    IFH_CHECK_ARG((((uintptr_t)d->x) & 7) == 0 && (((uintptr_t)d->out) & 7) == 0 && (((uintptr_t)d->wstream) & 15) == 0 &&
                  (((uintptr_t)d->bias) & 15) == 0 && d->x_bstride % 4 == 0 && d->out_bstride % 4 == 0);
    IFH_CHECK_ARG(d->slope > 0.0f && d->slope <= 1.0f && (int64_t)d->nbatch * d->t < (1ll << 31));
    SeqParams p;
    p.x = (const uint16_t *)d->x;
    p.x_bstride = d->x_bstride;
    p.wstream = (const uint16_t *)d->wstream;
    p.bias = d->bias;
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
    hipStream_t st = as_stream(stream);
    int rc = IFH_EINVAL;
    //                   <C, TAPS, TSEQ, NSEQ, NW, WGN, NT, MT, NH, HSEQ, NRING, UNITB>
#define SEQ_CASE(C_, K_, ...)                      \
    if (d->c == C_ && d->taps == K_) rc = launch_seq<C_, K_, __VA_ARGS__>(p, st);
    // C = 64: one 768-row sequence per workgroup, eight waves of 96 rows x 64 channels, full steps
    SEQ_CASE(64, 11, 768, 1, 8, 1, 4, 6, 1, false, 3, 8192)
    SEQ_CASE(64, 7, 768, 1, 8, 1, 4, 6, 1, false, 4, 8192)
    // (taps = 3 at C = 64 stays with ifh_resblock_chain_bf16: six k-steps per convolution unroll into straight-line code that hipcc
    // fills with spills -- some of them right behind asm reads, tools/lint_asm_loads.py -- and it measured no faster, 350 against 287 us)
    // C = 128: two 192-row sequences per workgroup, a wave = 96 rows x 64 channels
    SEQ_CASE(128, 11, 192, 2, 8, 2, 4, 6, 1, false, 3, 8192)
    SEQ_CASE(128, 7, 192, 2, 8, 2, 4, 6, 1, false, 4, 8192)
    SEQ_CASE(128, 3, 192, 2, 8, 2, 4, 6, 1, false, 4, 8192)
    // C = 256: two 48-row sequences per workgroup, four waves (512 registers) of both sequences x 64 channels, a half-step per
    // sequence; a k-step of weights is one 16 KB unit
    constexpr int nh256 = 2;     // fixed by measurement (profiles/NOTES.md): half-steps (a sequence each) or full steps
    if (nh256 == 2) {
        SEQ_CASE(256, 11, 48, 2, 4, 4, 4, 6, 2, true, 4, 16384)
        SEQ_CASE(256, 7, 48, 2, 4, 4, 4, 6, 2, true, 4, 16384)
        SEQ_CASE(256, 3, 48, 2, 4, 4, 4, 6, 2, true, 4, 16384)
    } else {
        SEQ_CASE(256, 11, 48, 2, 4, 4, 4, 6, 1, true, 4, 16384)
        SEQ_CASE(256, 7, 48, 2, 4, 4, 4, 6, 1, true, 4, 16384)
        SEQ_CASE(256, 3, 48, 2, 4, 4, 4, 6, 1, true, 4, 16384)
    }
#undef SEQ_CASE
    if (rc != IFH_OK) return rc;
    IFH_LAUNCH_CHECK("resblock_seq_bf16");
    return IFH_OK;
}
