// step.hip -- the RESIDENT decode step: one launch per SpeechT5 decoder step instead of ~55 dependent ones.
//
// The reference loops its decoder 16 steps per infer() call (HelloSippyTTSRT/HelloSippyRTPipe.py:196-229); rounds 1-3 ran a step
// as a chain of ~55 small launches (4 prenet GEMMs, 6 layers x {qkv, self-attention, wo, cross-q, cross-attention, cross-wo, fc1,
// fc2}, feat, prob, stop rule).  Inside the pipelined serving cycle that chain is the bottleneck: every launch has to find CU
// slots beside the resident encoder / vocoder workgroups (profiles/NOTES.md, round 4: median launch 12 us, mean 19-41 us).
//
// Every operation of the step is ROW-LOCAL (a row's outputs depend on that row's inputs and the weights only), so the step needs no
// grid-wide synchronisation at all: the rows are cut into blocks of RB = 32, and a CLUSTER of `cw` workgroups walks one row block
// through all phases, synchronising only with itself -- one agent-scope arrival counter per cluster, (phases - 1) x cw arrivals
// per step.  Clusters never wait for each other.  Workgroup b belongs to cluster (b % 8) + 8 * (b / (8 * cw)): blocks b and b + 8
// land on the same XCD under the round-robin dispatch (MI355X_MICROARCH.md, workgroup dispatch), so a cluster's hand-offs stay
// inside one L2 when that holds; correctness does not depend on it (below).
//
// Hand-off between the workgroups of a cluster, as the micro-architecture guide prescribes for producers and consumers inside
// one launch: every handed-off byte is stored write-through (sc1: IFH_EPI_SC1 in igemm.h, buffer stores here) and loaded with
// sc1 loads (never served by a CU's L1); every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a
// barrier, one lane adds to the cluster counter; the consumer's one lane polls the counter with sc1 loads, the workgroup meets at
// a barrier, then loads.  LayerNorm statistics keep their integer atomics (agent scope).  Every spin is bounded (2 s of the
// 100 MHz clock): a cluster that cannot become co-resident sets the context's error word instead of hanging the GPU.
//
// ARITHMETIC: a phase is the IgemmParams / attention arguments the launch chain would have used, recorded on the host
// (ifh_step_record_begin ... ifh_step_record_end: ifh_conv_bf16, ifh_attn_decode_bf16 and ifh_tts_stop_advance_rows append to the
// table instead of launching).  GEMM phases accumulate K in the SAME chains as k_gemm_skinny / k_gemm_dec (2 below K = 2048, 4 from
// there, contiguous k ranges, added in chain order) and finish in the SAME epilogue (ln_epi4 / igemm_store4_fast); attention
// phases run attn_row_update over the keys in the order of k_attn_decode<4> (self-attention: the four waves' key sets one after the
// other in ONE wave, merged in wave order) and k_attn_decode<1> (cross-attention).  Same bits as the launch chain:
// tests/test_step_resident_gpu.py.
#define IFH_EPI_SC1 1
#include <vector>

#include "attn_core.h"
#include "igemm.h"

namespace ifh {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

enum { PH_GEMM = 0, PH_ATTN = 1, PH_STOP = 2 };

struct AttnArgs {        // ifh_attn_decode_bf16
    const uint16_t *q, *k, *v;
    uint16_t *ao;
    int64_t q_bs, kv_bs, kv_ts, o_bs;
    const int32_t *key_len, *dyn_len;
    int S, dyn_add, nheads, nw;
};
struct StopArgs {        // ifh_tts_stop_advance_rows: stop rule + position advance + the rows' LayerNorm statistics cleared
    const float *logits;
    int64_t *ends_at;
    int32_t *pos;
    const uint8_t *active;
    const int32_t *minmax;
    float thr;
    int ends_inc, ld;
    uint4 *zbuf;
    int zslots, zrows;
};
struct StepPhase {
    int kind, ksplit;
    IgemmParams g;
    AttnArgs a;
    StopArgs s;
};
// the phase table is read through the constant address space: scalar loads into SGPRs whatever the kernel stores elsewhere
typedef const StepPhase __attribute__((address_space(4))) *PhaseTab;
template <typename T>
__device__ __forceinline__ T ld_const(const T __attribute__((address_space(4))) *src)
{
    T v;
    __builtin_memcpy(&v, src, sizeof(T));
    return v;
}

#ifndef IFH_STEP_UBL
#define IFH_STEP_UBL 12      // k-steps of weights in flight per wave, activation image in LDS
#endif
#ifndef IFH_STEP_UBD
#define IFH_STEP_UBD 6       // ... activations straight from memory (K beyond the image)
#endif
#ifndef IFH_STEP_SW
#define IFH_STEP_SW 8        // (16 waves = 128 VGPRs per lane: the attention phase spilled, and its four-wave emulation came out wrong)
#endif
constexpr int SW = IFH_STEP_SW;            // waves per workgroup
constexpr int KCAP = 768, LDK = KCAP + 8;  // activation image in LDS: up to 768 of K per row, rows 16 bytes apart modulo 128
constexpr int CTR_STRIDE = 16;             // counters / epochs of different clusters on lines of their own (u64 units)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 ld16_sc1(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, uint4 v)
{
    const u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, byte_off, 0, 16);
}
__device__ __forceinline__ uint2 ld8_sc1(const void *ptr)
{
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

// ---- cluster synchronisation -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cluster_arrive(unsigned long long *ctr)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores (and atomics) have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cluster_wait(const unsigned long long *ctr, unsigned long long target, int *err)
{
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {      // 2 s: the cluster is not co-resident
                atomicExch(err, 1);
                break;
            }
        }
    }
    __syncthreads();
}

// ---- GEMM phase ----------------------------------------------------------------------------------------------------------------
struct RowAddr {
    int off;       // element offset of the row in p.x
    bool ok;
};
__device__ __forceinline__ RowAddr x_row(const IgemmParams &p, int m, int M)
{
    RowAddr r;
    r.ok = m < M;
    const int mm = r.ok ? m : 0;
    const int b = p.T_out == 1 ? mm : mm / p.T_out, t = mm - b * p.T_out;
    r.off = (int)((int64_t)b * p.x_bstride + (int64_t)t * p.lda);
    return r;
}

// one accumulation chain (k-steps ks0 .. ks1 - 1) of one 16-column tile over MT row tiles.  The activation fragments come from the
// LDS image (As) or, for K beyond its capacity, straight from memory (DIRECT).
template <int MT, int UB, bool DIRECT>
__device__ __forceinline__ void gemm_chain(const IgemmParams &p, const uint16_t *As, __amdgpu_buffer_rsrc_t xr, int r0, int M, int n0,
                                           int ks0, int ks1, int fr, int fg, f32x4 (&acc)[MT])
{
    const int nrow = n0 + fr;
    const bool wok = nrow < p.N;
    const uint16_t *wrow = p.w + (int64_t)(wok ? nrow : 0) * p.K + fg * 8;
    int xoff[MT];
    bool xok[MT];
#pragma unroll
    for (int j = 0; j < MT; j++) {
        const RowAddr ra = x_row(p, r0 + j * 16 + fr, M);
        xoff[j] = (ra.off + fg * 8) * 2;
        xok[j] = ra.ok;
    }
    for (int ks = ks0; ks < ks1; ks += UB) {
        uint4 wv[UB], xv[DIRECT ? UB * MT : 1];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int kk = ks + u;
            const bool kok = kk < ks1 && (kk * 32 + fg * 8) < p.K;      // K % 8 == 0
            wv[u] = make_uint4(0, 0, 0, 0);
            if (wok && kok) wv[u] = *reinterpret_cast<const uint4 *>(wrow + kk * 32);
            if (DIRECT) {
#pragma unroll
                for (int j = 0; j < MT; j++) {
                    xv[u * MT + j] = make_uint4(0, 0, 0, 0);
                    if (xok[j] && kok) xv[u * MT + j] = ld16_sc1(xr, xoff[j] + kk * 64);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int kk = ks + u;
            if (kk < ks1) {
                const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wv[u]);
#pragma unroll
                for (int j = 0; j < MT; j++) {
                    bf16x8_t xf;
                    if (DIRECT)
                        xf = __builtin_bit_cast(bf16x8_t, xv[u * MT + j]);
                    else
                        xf = *reinterpret_cast<const bf16x8_t *>(&As[(j * 16 + fr) * LDK + kk * 32 + fg * 8]);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[j], 0, 0, 0);
                }
            }
        }
    }
}

// the epilogue of one 16 x 16 output tile (column tile n0, rows mrow0 ..): operands as k_gemm_skinny / k_gemm_dec fetch them,
// with sc1 loads for what other workgroups of this launch have produced (row statistics, residual rows)
__device__ __forceinline__ void gemm_epilogue(const IgemmParams &p, int M, int n0, int mrow0, f32x4 s, int fr, int fg)
{
    const int em = mrow0 + fr;
    const bool exok = em < M;
    const int edyn = dyn_value(p, exok ? em : 0);
    const int n = n0 + 4 * fg;
    const bool ln_mode = p.aln_stats || p.rln_stats || p.stats_out;
    if (ln_mode) {
        longlong2 st_a = make_longlong2(0, 0), st_r = make_longlong2(0, 0);
        if (exok) {
            if (p.aln_stats) {
                const uint4 t = ld16_sc1(mk_rsrc(p.aln_stats), em * 16);
                st_a = make_longlong2((long long)((unsigned long long)t.x | ((unsigned long long)t.y << 32)),
                                      (long long)((unsigned long long)t.z | ((unsigned long long)t.w << 32)));
            }
            if (p.rln_stats) {
                const uint4 t = ld16_sc1(mk_rsrc(p.rln_stats), em * 16);
                st_r = make_longlong2((long long)((unsigned long long)t.x | ((unsigned long long)t.y << 32)),
                                      (long long)((unsigned long long)t.z | ((unsigned long long)t.w << 32)));
            }
        }
        float4 pc1 = make_float4(0.f, 0.f, 0.f, 0.f), pbias = pc1, pgam = pc1, pbeta = pc1;
        uint2 presid = make_uint2(0, 0);
        if (n < p.N) {
            if (p.aln_stats && !p.ln_rms) pc1 = *reinterpret_cast<const float4 *>(p.aln_c1 + n);
            if (p.bias) pbias = *reinterpret_cast<const float4 *>(p.bias + n);
            if (p.resid && p.rln_stats) {
                pgam = *reinterpret_cast<const float4 *>(p.rln_gamma + n);
                pbeta = *reinterpret_cast<const float4 *>(p.rln_beta + n);
            }
            if (p.resid && exok) presid = ld8_sc1(p.resid + epi_row(p, em, n, edyn).rbase + n);
        }
        ln_epi4(p, em, n, exok, s, edyn, ln_row(p, st_a, st_r), pc1, pbias, pgam, pbeta, presid, fg);
    } else if (exok && n < p.N) {
        uint2 rpre = make_uint2(0, 0);
        if (p.resid) rpre = ld8_sc1(p.resid + epi_row(p, em, n, edyn).rbase + n);
        (void)igemm_store4_fast<true>(p, em, n, s, edyn, rpre);
    }
}

template <int MT>
__device__ __forceinline__ void phase_gemm(const IgemmParams &p, int ksplit, int r0, int member, int cw, uint16_t *As, f32x4 *red)
{
    constexpr int RB = 16 * MT;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int M = p.nbatch * p.T_out;
    const int ntile = (p.N + 15) / 16;
    const int t_lo = (int)((int64_t)member * ntile / cw), t_hi = (int)((int64_t)(member + 1) * ntile / cw);
    const int nk = (p.K + 31) / 32;
    const int per = (nk + ksplit - 1) / ksplit;
    const bool direct = p.K > KCAP;
    const __amdgpu_buffer_rsrc_t xr = mk_rsrc(p.x);
    if (!direct && t_hi > t_lo) {
        // the row block's activations -> LDS (zeros for rows beyond M and for k beyond K)
        const int vpr = nk * 4;                               // 16-byte vectors per row
        for (int i = tid; i < RB * vpr; i += SW * 64) {
            const int row = i / vpr, kv = i - row * vpr;
            const RowAddr ra = x_row(p, r0 + row, M);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ra.ok && kv * 8 < p.K) v = ld16_sc1(xr, (ra.off + kv * 8) * 2);
            *reinterpret_cast<uint4 *>(&As[row * LDK + kv * 8]) = v;
        }
    }
    __syncthreads();
    const int ntask = (t_hi - t_lo) * ksplit;
    for (int rb = 0; rb < ntask; rb += SW) {
        const int task = rb + wid;
        if (task < ntask) {
            const int n0 = (t_lo + task / ksplit) * 16, c = task % ksplit;
            const int ks0 = c * per, ks1 = min(nk, ks0 + per);
            f32x4 acc[MT];
#pragma unroll
            for (int j = 0; j < MT; j++) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (direct)
                gemm_chain<MT, IFH_STEP_UBD, true>(p, As, xr, r0, M, n0, ks0, ks1, fr, fg, acc);
            else
                gemm_chain<MT, IFH_STEP_UBL, false>(p, As, xr, r0, M, n0, ks0, ks1, fr, fg, acc);
#pragma unroll
            for (int j = 0; j < MT; j++) red[(wid * MT + j) * 64 + lane] = acc[j];
        }
        __syncthreads();
        // tiles finished in this round: their chains are added in chain order, then the shared epilogue
        const int tr = min(SW, ntask - rb) / ksplit;
        if (wid < tr * MT) {
            const int tl = wid / MT, j = wid - tl * MT;
            f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < ksplit; c++) s += red[((tl * ksplit + c) * MT + j) * 64 + lane];
            gemm_epilogue(p, M, (t_lo + rb / ksplit + tl) * 16, r0 + j * 16, s, fr, fg);
        }
        __syncthreads();
    }
}

// ---- attention phase -----------------------------------------------------------------------------------------------------------
// the keys wave `w` of an NW-wave k_attn_decode workgroup walks, for the query row held in q2: online softmax state (m, l, o) merged
// over the wave's 8 key groups -- what that wave contributes to the workgroup's result
template <int NW>
__device__ __forceinline__ void attn_partial(int w, int klen, const f32x2 (&q2)[4], __amdgpu_buffer_rsrc_t kr, __amdgpu_buffer_rsrc_t vr,
                                             const uint16_t *kp, const uint16_t *vp_, int kvoff, int kv_ts2, int g, float &m, float &l,
                                             float (&o)[8])
{
    constexpr int KU = 4;
    m = -1e30f;
    l = 0.0f;
    f32x2 o2[4];
#pragma unroll
    for (int e = 0; e < 4; e++) o2[e] = (f32x2){0.0f, 0.0f};
    for (int key0 = w * 8 + g; key0 < klen; key0 += 8 * NW * KU) {
        uint4 kk[KU], vv[KU];
        bool valid[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int key = key0 + u * 8 * NW;
            valid[u] = key < klen;
            kk[u] = make_uint4(0, 0, 0, 0);
            vv[u] = make_uint4(0, 0, 0, 0);
            if (valid[u]) {
#ifdef IFH_STEP_ATTN_PLAIN
                kk[u] = ld_stream16(kp + (kvoff + key * kv_ts2) / 2);
                vv[u] = ld_stream16(vp_ + (kvoff + key * kv_ts2) / 2);
#else
                kk[u] = ld16_sc1(kr, kvoff + key * kv_ts2);
                vv[u] = ld16_sc1(vr, kvoff + key * kv_ts2);
#endif
            }
        }
        f32x2 klo[KU / 2][4], khi[KU / 2][4], vp[KU][4];
        attn_unpack<KU>(kk, vv, klo, khi, vp);
        attn_row_update<KU>(q2, klo, khi, vp, valid, m, l, o2);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        o[2 * e] = o2[e].x;
        o[2 * e + 1] = o2[e].y;
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
        const float mn = fmaxf(m, m2);
        const float a = __expf(m - mn), a2 = __expf(m2 - mn);
        l = l * a + l2 * a2;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = o[i] * a + __shfl_xor(o[i], off, 64) * a2;
        m = mn;
    }
}

template <int NW>
__device__ __forceinline__ void attn_unit(const AttnArgs &P, int row, int h, int lane)
{
    const int c = lane & 7, g = lane >> 3;
    const int klen = P.key_len ? P.key_len[row] + P.dyn_add : (P.dyn_len ? P.dyn_len[0] + P.dyn_add : P.S);
    f32x2 q2[4];
    {
        const uint4 t = ld16_sc1(mk_rsrc(P.q), (int)(((int64_t)row * P.q_bs + h * 64 + 8 * c) * 2));
        const uint32_t *u = reinterpret_cast<const uint32_t *>(&t);
#pragma unroll
        for (int e = 0; e < 4; e++) q2[e] = (f32x2){__uint_as_float(u[e] << 16), __uint_as_float(u[e] & 0xffff0000u)};
    }
    const __amdgpu_buffer_rsrc_t kr = mk_rsrc(P.k + (int64_t)row * P.kv_bs), vr = mk_rsrc(P.v + (int64_t)row * P.kv_bs);
    const uint16_t *kp = P.k + (int64_t)row * P.kv_bs, *vp_ = P.v + (int64_t)row * P.kv_bs;
    const int kvoff = (h * 64 + 8 * c) * 2, kv_ts2 = (int)(P.kv_ts * 2);
    float m, l, o[8];
    attn_partial<NW>(0, klen, q2, kr, vr, kp, vp_, kvoff, kv_ts2, g, m, l, o);
    if (NW > 1) {
#ifdef IFH_STEP_ATTN_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int w = 1; w < NW; w++) {
            float m2, l2, o2[8];
            attn_partial<NW>(w, klen, q2, kr, vr, kp, vp_, kvoff, kv_ts2, g, m2, l2, o2);
            const float mn = fmaxf(m, m2);
            const float a = __expf(m - mn), a2 = __expf(m2 - mn);
            l = l * a + l2 * a2;
#pragma unroll
            for (int i = 0; i < 8; i++) o[i] = o[i] * a + o2[i] * a2;
            m = mn;
        }
    }
    if (g == 0) {
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        uint4 pk;
        pk.x = pack2(o[0] * inv, o[1] * inv);
        pk.y = pack2(o[2] * inv, o[3] * inv);
        pk.z = pack2(o[4] * inv, o[5] * inv);
        pk.w = pack2(o[6] * inv, o[7] * inv);
        st16_sc1(mk_rsrc(P.ao), (int)(((int64_t)row * P.o_bs + h * 64 + 8 * c) * 2), pk);
    }
}

__device__ __forceinline__ void phase_attn(const AttnArgs &P, int r0, int rows_here, int member, int cw)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int units = rows_here * P.nheads;
    for (int u = member * SW + wid; u < units; u += cw * SW) {
        const int row = __builtin_amdgcn_readfirstlane(r0 + u / P.nheads), h = __builtin_amdgcn_readfirstlane(u % P.nheads);
        if (P.nw == 4)
            attn_unit<4>(P, row, h, lane);
        else
            attn_unit<1>(P, row, h, lane);
    }
}

// ---- stop rule, position advance, the rows' statistics cleared for the next step (k_tts_stop_advance_rows) ----
__device__ __forceinline__ void phase_stop(const StopArgs &P, int r0, int rows_here, int member)
{
    if (member != 0) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < P.zslots * rows_here; i += SW * 64) {
        const int s = i / rows_here, r = i - s * rows_here;
        P.zbuf[(int64_t)s * P.zrows + r0 + r] = make_uint4(0, 0, 0, 0);
    }
    if (tid < rows_here) {
        const int b = r0 + tid;
        if (P.active[b]) {
            const int idx = P.pos[b], minlen = P.minmax[2 * b], maxlen = P.minmax[2 * b + 1];
            const uint2 lg = ld8_sc1(P.logits + (int64_t)P.ld * b);
            const float p0 = 1.0f / (1.0f + expf(-__uint_as_float(lg.x))), p1 = 1.0f / (1.0f + expf(-__uint_as_float(lg.y)));
            const bool hit = (P.ends_at[b] < 0) && (minlen <= idx) && ((p0 >= P.thr) || (p1 >= P.thr) || (maxlen <= idx));
            if (hit) P.ends_at[b] = idx + P.ends_inc;
            P.pos[b] = idx + 1;
        }
    }
}

template <int MT>
__global__ __launch_bounds__(SW * 64) void k_step_resident(const StepPhase *__restrict__ tab, int nphase, int nrows, int cw,
                                                           unsigned long long *__restrict__ ctrs, unsigned long long *__restrict__ epochs,
                                                           int *__restrict__ err, int *__restrict__ dbg_xcc,
                                                           unsigned long long *__restrict__ prof)
{
    constexpr int RB = 16 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t *As = reinterpret_cast<uint16_t *>(lds_raw);
    f32x4 *red = reinterpret_cast<f32x4 *>(lds_raw + RB * LDK * 2);
    const int b = blockIdx.x;
    const int j = b >> 3, kc = j / cw, member = j - kc * cw;
    const int cluster = (b & 7) + 8 * kc;
    if (dbg_xcc && threadIdx.x == 0) dbg_xcc[b] = __builtin_amdgcn_s_getreg(63508);      // HW_REG_XCC_ID
    const int r0 = cluster * RB;
    if (r0 >= nrows) return;
    const int rows_here = min(RB, nrows - r0);
    unsigned long long *ctr = ctrs + (int64_t)cluster * CTR_STRIDE;
    // the counter's value when this launch began: left by the previous launch (member 0 stores it once every member has read it,
    // i.e. after the first wait)
    const unsigned long long e0 = __hip_atomic_load(epochs + (int64_t)cluster * CTR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const PhaseTab ctab = (PhaseTab)tab;
    // prof (debug): 100 MHz ticks cluster 0 / member 0 spent per phase -- [2 ph] waiting for the cluster, [2 ph + 1] in the phase
    const bool stamp = prof && cluster == 0 && member == 0 && threadIdx.x == 0;
    unsigned long long tprev = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
    for (int ph = 0; ph < nphase; ph++) {
        const int kind = ctab[ph].kind;
        if (ph > 0) {
            cluster_wait(ctr, e0 + (unsigned long long)ph * cw, err);
            if (stamp) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                prof[2 * ph] += now - tprev;
                tprev = now;
            }
            if (ph == 1 && member == 0 && threadIdx.x == 0) epochs[(int64_t)cluster * CTR_STRIDE] = e0 + (unsigned long long)(nphase - 1) * cw;
        }
        if (kind == PH_GEMM) {
            const IgemmParams p = ld_const(&ctab[ph].g);
            phase_gemm<MT>(p, ctab[ph].ksplit, r0, member, cw, As, red);
        } else if (kind == PH_ATTN) {
            const AttnArgs a = ld_const(&ctab[ph].a);
            phase_attn(a, r0, rows_here, member, cw);
        } else {
            const StopArgs sa = ld_const(&ctab[ph].s);
            phase_stop(sa, r0, rows_here, member);
        }
        if (ph + 1 < nphase) cluster_arrive(ctr);
        if (stamp) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            prof[2 * ph + 1] += now - tprev;
            tprev = now;
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
struct StepRecorder {
    bool on = false;
    int stat_rows = 0, rows = -1;
    std::vector<StepPhase> ph;
};
static thread_local StepRecorder g_rec;

struct StepProg {
    StepPhase *tab = nullptr;
    int nphase = 0, nrows = 0;
};
struct StepCtx {
    unsigned long long *ctrs = nullptr, *epochs = nullptr;
    int *err = nullptr, *dbg = nullptr;
    unsigned long long *prof = nullptr;      // [256]: per-phase ticks of cluster 0 (debug)
    int max_clusters = 0;
};

bool step_recording() { return g_rec.on; }

static int rec_rows(int rows)
{
    if (g_rec.rows < 0) g_rec.rows = rows;
    if (g_rec.rows != rows) return fail(IFH_EINVAL, "step record: every phase of a resident step covers the same rows");
    return IFH_OK;
}

int step_record_gemm(const IgemmParams &p)
{
    const int64_t M = (int64_t)p.nbatch * p.T_out;
    const bool ln_mode = p.aln_stats || p.rln_stats || p.stats_out;
    if (!(p.taps == 1 && p.stride == 1 && p.pad == 0 && p.pre_slope == 1.0f && !p.accumulate && p.zt_cout == 0 && p.K % 8 == 0 &&
          M >= 1 && M <= 1024 && (ln_mode || p.fast_epi) && (!ln_mode || p.N % 16 == 0)))
        return fail(IFH_EINVAL, "step record: not a decode-step GEMM the resident kernel takes");
    // 32-bit byte offsets of the buffer loads
    const int64_t span = ((int64_t)(p.nbatch - 1) * p.x_bstride + (int64_t)(p.T_out - 1) * p.lda + p.K) * 2;
    if (span >= (1ll << 31) || M * 16 >= (1ll << 31)) return fail(IFH_EINVAL, "step record: activation span");
    if (int rc = rec_rows((int)M)) return rc;
    StepPhase s{};
    s.kind = PH_GEMM;
    s.ksplit = p.K >= 2048 ? 4 : 2;       // the K split of k_gemm_skinny / k_gemm_dec (nn.hip): a function of K alone
    s.g = p;
    g_rec.ph.push_back(s);
    return IFH_OK;
}

int step_record_attn(const void *q, int64_t q_bs, const void *k, const void *v, int64_t kv_bs, int64_t kv_ts, void *out, int64_t o_bs,
                     const int32_t *key_len, int max_keys, int nbatch, int nheads, const int32_t *dyn_len, int dyn_add, int kv_group)
{
    if (kv_group != 1) return fail(IFH_EINVAL, "step record: shared cache rows are not a resident-step phase");
    if (((int64_t)nbatch * q_bs + nheads * 64) * 2 >= (1ll << 31) || ((int64_t)nbatch * o_bs + nheads * 64) * 2 >= (1ll << 31) ||
        ((int64_t)max_keys * kv_ts + nheads * 64) * 2 >= (1ll << 31))
        return fail(IFH_EINVAL, "step record: attention spans");
    if (int rc = rec_rows(nbatch)) return rc;
    StepPhase s{};
    s.kind = PH_ATTN;
    s.a.q = (const uint16_t *)q;
    s.a.k = (const uint16_t *)k;
    s.a.v = (const uint16_t *)v;
    s.a.ao = (uint16_t *)out;
    s.a.q_bs = q_bs;
    s.a.kv_bs = kv_bs;
    s.a.kv_ts = kv_ts;
    s.a.o_bs = o_bs;
    s.a.key_len = key_len;
    s.a.dyn_len = dyn_len;
    s.a.S = max_keys;
    s.a.dyn_add = dyn_add;
    s.a.nheads = nheads;
    s.a.nw = (max_keys > 256 || dyn_len || (key_len && dyn_add)) ? 4 : 1;       // attn_decode_launch's choice (attn.hip)
    g_rec.ph.push_back(s);
    return IFH_OK;
}

int step_record_stop(const float *prob_logits, int64_t *ends_at, int n, float threshold, int ends_inc, int32_t *pos,
                     const uint8_t *active, const int32_t *minmax, int logits_ld, void *zero_buf, int64_t zero_bytes)
{
    if (g_rec.stat_rows <= 0 || zero_bytes % ((int64_t)g_rec.stat_rows * 16) != 0 || (((uintptr_t)prob_logits) & 7) || logits_ld % 2)
        return fail(IFH_EINVAL, "step record: statistics layout");
    if (int rc = rec_rows(n)) return rc;
    StepPhase s{};
    s.kind = PH_STOP;
    s.s.logits = prob_logits;
    s.s.ends_at = ends_at;
    s.s.pos = pos;
    s.s.active = active;
    s.s.minmax = minmax;
    s.s.thr = threshold;
    s.s.ends_inc = ends_inc;
    s.s.ld = logits_ld;
    s.s.zbuf = (uint4 *)zero_buf;
    s.s.zrows = g_rec.stat_rows;
    s.s.zslots = (int)(zero_bytes / ((int64_t)g_rec.stat_rows * 16));
    g_rec.ph.push_back(s);
    return IFH_OK;
}

}  // namespace ifh

using namespace ifh;

extern "C" int ifh_step_record_begin(int stat_rows)
{
    IFH_CHECK_ARG(!g_rec.on && stat_rows > 0);
    g_rec.on = true;
    g_rec.stat_rows = stat_rows;
    g_rec.rows = -1;
    g_rec.ph.clear();
    return IFH_OK;
}

extern "C" int ifh_step_record_abort(void)
{
    g_rec.on = false;
    g_rec.ph.clear();
    return IFH_OK;
}

extern "C" int ifh_step_record_end(ifh_step_prog_t *prog_out, int *nphase_out)
{
    IFH_CHECK_ARG(g_rec.on && prog_out);
    g_rec.on = false;
    IFH_CHECK_ARG(!g_rec.ph.empty() && g_rec.rows > 0);
    StepProg *pr = new StepProg;
    pr->nphase = (int)g_rec.ph.size();
    pr->nrows = g_rec.rows;
    if (int rc = check_hip(hipMalloc((void **)&pr->tab, sizeof(StepPhase) * g_rec.ph.size()), "step prog alloc")) {
        delete pr;
        return rc;
    }
    if (int rc = check_hip(hipMemcpy(pr->tab, g_rec.ph.data(), sizeof(StepPhase) * g_rec.ph.size(), hipMemcpyHostToDevice), "step prog upload")) {
        (void)hipFree(pr->tab);
        delete pr;
        return rc;
    }
    g_rec.ph.clear();
    if (nphase_out) *nphase_out = pr->nphase;
    *prog_out = pr;
    return IFH_OK;
}

extern "C" int ifh_step_prog_destroy(ifh_step_prog_t prog)
{
    StepProg *pr = reinterpret_cast<StepProg *>(prog);
    if (!pr) return IFH_OK;
    (void)hipFree(pr->tab);
    delete pr;
    return IFH_OK;
}

extern "C" int ifh_step_ctx_create(int max_rows, ifh_step_ctx_t *ctx_out)
{
    IFH_CHECK_ARG(ctx_out && max_rows > 0 && max_rows <= 1024);
    StepCtx *c = new StepCtx;
    c->max_clusters = (max_rows + 15) / 16 + 8;
    const size_t nb = sizeof(unsigned long long) * CTR_STRIDE * c->max_clusters;
    hipError_t e = hipMalloc((void **)&c->ctrs, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&c->epochs, nb);
    if (e == hipSuccess) e = hipMalloc((void **)&c->err, 256);
    if (e == hipSuccess) e = hipMalloc((void **)&c->dbg, sizeof(int) * 4096);
    if (e == hipSuccess) e = hipMemset(c->ctrs, 0, nb);
    if (e == hipSuccess) e = hipMemset(c->epochs, 0, nb);
    if (e == hipSuccess) e = hipMemset(c->err, 0, 256);
    if (e == hipSuccess) e = hipMemset(c->dbg, 0xff, sizeof(int) * 4096);
    if (e == hipSuccess) e = hipMalloc((void **)&c->prof, sizeof(unsigned long long) * 256);
    if (e == hipSuccess) e = hipMemset(c->prof, 0, sizeof(unsigned long long) * 256);
    if (e != hipSuccess) {
        (void)hipFree(c->prof);
        (void)hipFree(c->ctrs);
        (void)hipFree(c->epochs);
        (void)hipFree(c->err);
        (void)hipFree(c->dbg);
        delete c;
        return check_hip(e, "step ctx alloc");
    }
    *ctx_out = c;
    return IFH_OK;
}

extern "C" int ifh_step_ctx_destroy(ifh_step_ctx_t ctx)
{
    StepCtx *c = reinterpret_cast<StepCtx *>(ctx);
    if (!c) return IFH_OK;
    (void)hipFree(c->ctrs);
    (void)hipFree(c->epochs);
    (void)hipFree(c->err);
    (void)hipFree(c->dbg);
    (void)hipFree(c->prof);
    delete c;
    return IFH_OK;
}

// debug: the per-phase 100 MHz ticks accumulated by launches with (debug & 2) -- out[2 ph] waiting, out[2 ph + 1] working -- then cleared
extern "C" int ifh_step_ctx_prof(ifh_step_ctx_t ctx, unsigned long long *out256)
{
    StepCtx *c = reinterpret_cast<StepCtx *>(ctx);
    IFH_CHECK_ARG(c && out256);
    if (int rc = check_hip(hipDeviceSynchronize(), "step ctx sync")) return rc;
    if (int rc = check_hip(hipMemcpy(out256, c->prof, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost), "step ctx prof")) return rc;
    return check_hip(hipMemset(c->prof, 0, sizeof(unsigned long long) * 256), "step ctx prof clear");
}

// the error word (1: a cluster wait ran into its 2 s bound) and, when xcc_out is given, the XCC id every workgroup of the last
// launch ran on (n_xcc ints, -1 where no workgroup wrote).  Synchronises the device.  An error clears the counters.
extern "C" int ifh_step_ctx_status(ifh_step_ctx_t ctx, int *err_out, int *xcc_out, int n_xcc)
{
    StepCtx *c = reinterpret_cast<StepCtx *>(ctx);
    IFH_CHECK_ARG(c && err_out && n_xcc >= 0 && n_xcc <= 4096);
    if (int rc = check_hip(hipDeviceSynchronize(), "step ctx sync")) return rc;
    if (int rc = check_hip(hipMemcpy(err_out, c->err, sizeof(int), hipMemcpyDeviceToHost), "step ctx err")) return rc;
    if (xcc_out && n_xcc)
        if (int rc = check_hip(hipMemcpy(xcc_out, c->dbg, sizeof(int) * n_xcc, hipMemcpyDeviceToHost), "step ctx xcc")) return rc;
    if (*err_out) {
        const size_t nb = sizeof(unsigned long long) * CTR_STRIDE * c->max_clusters;
        (void)hipMemset(c->ctrs, 0, nb);
        (void)hipMemset(c->epochs, 0, nb);
        (void)hipMemset(c->err, 0, 256);
    }
    return IFH_OK;
}

extern "C" int ifh_step_run(ifh_step_prog_t prog, ifh_step_ctx_t ctx, int cw, int debug, ifh_stream_t stream)
{
    StepProg *pr = reinterpret_cast<StepProg *>(prog);
    StepCtx *c = reinterpret_cast<StepCtx *>(ctx);
    IFH_CHECK_ARG(pr && c && cw >= 1 && cw <= 64 && pr->nphase <= 128);
    constexpr int MT = 2, RB = 16 * MT;
    const int nclusters = (pr->nrows + RB - 1) / RB;
    const int groups = (nclusters + 7) / 8;
    IFH_CHECK_ARG(groups * 8 <= c->max_clusters && groups * 8 * cw <= 4096);
    constexpr size_t lds = (size_t)RB * LDK * 2 + (size_t)SW * MT * 64 * sizeof(f32x4);
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_step_resident<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return check_hip(e, "step_resident lds attr");
        attr_once.done(attr_dev);
    }
    hipLaunchKernelGGL(k_step_resident<MT>, dim3(groups * 8 * cw), dim3(SW * 64), lds, as_stream(stream), (const StepPhase *)pr->tab,
                       pr->nphase, pr->nrows, cw, c->ctrs, c->epochs, c->err, (debug & 1) ? c->dbg : (int *)nullptr,
                       (debug & 2) ? c->prof : (unsigned long long *)nullptr);
    IFH_LAUNCH_CHECK("step_resident");
    return IFH_OK;
}
