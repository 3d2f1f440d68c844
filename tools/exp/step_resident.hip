// step.hip -- the RESIDENT decode step: one launch per SpeechT5 decoder step instead of ~55 dependent ones.
//
// The reference loops its decoder 16 steps per infer() call (HelloSippyTTSRT/HelloSippyRTPipe.py:196-229); rounds 1-3 ran a step
// as a chain of ~55 small launches (4 prenet GEMMs, 6 layers x {qkv, self-attention, wo, cross-q, cross-attention, cross-wo, fc1,
// fc2}, feat, prob, stop rule).  Inside the pipelined serving cycle that chain is the bottleneck: every launch has to find CU
// slots beside the resident encoder / vocoder workgroups (profiles/NOTES.md, round 4: median launch 12 us, mean 19-41 us).
//
// Every operation of the step is ROW-LOCAL (a row's outputs depend on that row's inputs and the weights only), so the step needs no
// grid-wide synchronisation at all: the rows are cut into blocks of RB = 32, and a CLUSTER of `cw` workgroups walks one row block
// through all phases, synchronising only with itself.  Clusters never wait for each other.  Workgroup b belongs to cluster
// (b % 8) + 8 * (b / (8 * cw)): blocks b and b + 8 land on the same XCD under the round-robin dispatch (MI355X_MICROARCH.md,
// workgroup dispatch), so a cluster normally shares ONE L2.
//
// Hand-off between the workgroups of a cluster (no atomics anywhere):
//   * arrival: every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one lane stores the
//     phase number into the workgroup's OWN flag word; the consumer's first wave polls the cw flags, one per lane, with sc1 loads,
//     then the workgroup meets at a barrier and loads;
//   * every load of handed-off bytes is an sc1 load (never served by a CU's L1);
//   * every store of handed-off bytes is write-through (sc1) -- unless the cluster has established, in its first wait, that all
//     its workgroups report the same XCC id: then stores are plain and the hand-offs stay in that XCD's L2 (a phase is 4-5
//     dependent round trips, and L2 answers in a third of the time memory does).  The placement is verified per launch, never
//     assumed;
//   * LayerNorm row statistics: each workgroup sums its tiles' fixed-point (2^16) sums in LDS and publishes them as ITS partial
//     for the row; the consumer adds the cw partials.  64-bit integer sums: the total is exactly what the launch chain's atomics
//     produce, and the statistics array itself is never touched (nothing to clear).
// Every spin is bounded (2 s of the 100 MHz clock): a cluster that cannot become co-resident sets the context's error word instead
// of hanging the GPU.
//
// ARITHMETIC: a phase is the IgemmParams / attention arguments the launch chain would have used, recorded on the host
// (ifh_step_record_begin ... ifh_step_record_end: ifh_conv_bf16, ifh_attn_decode_bf16 and ifh_tts_stop_advance_rows append to the
// table instead of launching).  GEMM phases accumulate K in the SAME chains as k_gemm_skinny / k_gemm_dec (2 below K = 2048, 4 from
// there, contiguous k ranges, added in chain order) and finish in the SAME epilogue (ln_epi4 / igemm_store4_fast); attention
// phases run attn_row_update over the keys in the order of k_attn_decode<4> (self-attention: the four waves' key sets in ONE wave,
// merged in wave order) and k_attn_decode<1> (cross-attention).  Same bits as the launch chain: tests/test_step_resident_gpu.py.
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
struct StopArgs {        // ifh_tts_stop_advance_rows: stop rule + position advance (the statistics it clears are not used here)
    const float *logits;
    int64_t *ends_at;
    int32_t *pos;
    const uint8_t *active;
    const int32_t *minmax;
    float thr;
    int ends_inc, ld;
    uint4 *zbuf;
    int64_t zbytes;
};
struct StepPhase {
    int kind, ksplit;
    int so_off, aln_off, rln_off;      // byte offsets of stats_out / aln_stats / rln_stats inside the statistics array (-1: none)
    int pad_;
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
#ifndef IFH_STEP_SW
#define IFH_STEP_SW 8        // (16 waves = 128 VGPRs per lane: the attention phase spilled, and its four-wave emulation came out wrong)
#endif
constexpr int SW = IFH_STEP_SW;            // waves per workgroup
constexpr int KCAP = 1536, LDK = KCAP + 8; // activation image in LDS: up to 1536 of K per row, rows 16 bytes apart modulo 128
constexpr int CWMAX = 32;                  // workgroups per cluster, at most
constexpr int LINE = 16;                   // u64 per 128-byte line: every flag / epoch word on a line of its own

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
// a load of handed-off bytes: sc1 (agent scope), never served by this CU's L1 -- in both hand-off modes: a mode switch around
// every load (the flag lives in LDS, so the compiler treats the branch as divergent and waits for each load where the two sides
// join) cost the loads' overlap, and non-temporal loads measured no faster than sc1 ones inside one XCD
__device__ __forceinline__ uint4 ld16_sc1(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// a store of handed-off bytes: plain inside one XCD's L2 (g_epi_plain), write-through otherwise
__device__ __forceinline__ void st16_out(__amdgpu_buffer_rsrc_t r, int byte_off, uint4 v)
{
    const u32x4 t = {v.x, v.y, v.z, v.w};
    if (g_epi_plain)
        __builtin_amdgcn_raw_buffer_store_b128(t, r, byte_off, 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b128(t, r, byte_off, 0, 16);
}
// K|V rows: read once per step and far larger than the caches -> non-temporal as well (attn_core.h: they must not evict the weights)
__device__ __forceinline__ uint4 ld16_sc1_nt(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16 | 2);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 ld8_sc1(const void *ptr)
{
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ longlong2 as_ll2(uint4 t)
{
    return make_longlong2((long long)((unsigned long long)t.x | ((unsigned long long)t.y << 32)),
                          (long long)((unsigned long long)t.z | ((unsigned long long)t.w << 32)));
}

// ---- cluster synchronisation -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cluster_arrive(unsigned long long *my_flag, unsigned long long value)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        if (g_epi_plain)
            *reinterpret_cast<volatile unsigned long long *>(my_flag) = value;
        else
            __hip_atomic_store(my_flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// every workgroup of the cluster has finished the phase numbered target - e0 - 1
__device__ __forceinline__ void cluster_wait(const unsigned long long *flags, int cw, unsigned long long target, int *err)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long *fp = flags + (int64_t)min(lane, cw - 1) * LINE;
        while (true) {
            const bool behind = lane < cw && __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target;
            if (!__any(behind)) break;
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {      // 2 s: the cluster is not co-resident
                if (lane == 0) atomicExch(err, 1);
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

template <int UB>
__device__ __forceinline__ void load_w(const uint16_t *wrow, bool wok, int K, int fg, int ks, int ks1, uint4 (&wv)[UB])
{
#pragma unroll
    for (int u = 0; u < UB; u++) {
        const int kk = ks + u;
        wv[u] = make_uint4(0, 0, 0, 0);
        if (wok && kk < ks1 && (kk * 32 + fg * 8) < K) wv[u] = *reinterpret_cast<const uint4 *>(wrow + kk * 32);      // K % 8 == 0
    }
}

// one accumulation chain (k-steps ks0 .. ks1 - 1) of one 16-column tile over MT row tiles, activation fragments from the LDS image;
// wv holds the chain's first UB k-steps of weights (requested ahead of the cluster wait: they depend on nothing)
template <int MT, int UB>
__device__ __forceinline__ void gemm_chain_lds(const uint16_t *As, const uint16_t *wrow, bool wok, int K, int ks0, int ks1, int fr, int fg,
                                               uint4 (&wv)[UB], f32x4 (&acc)[MT])
{
    int ks = ks0;
    while (true) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int kk = ks + u;
            if (kk < ks1) {
                const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wv[u]);
#pragma unroll
                for (int j = 0; j < MT; j++) {
                    const bf16x8_t xf = *reinterpret_cast<const bf16x8_t *>(&As[(j * 16 + fr) * LDK + kk * 32 + fg * 8]);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[j], 0, 0, 0);
                }
            }
        }
        ks += UB;
        if (ks >= ks1) break;
        load_w<UB>(wrow, wok, K, fg, ks, ks1, wv);
    }
}

// the cw workgroups' partial row statistics of row em, added up (integer sums: any order gives the launch chain's total)
__device__ __forceinline__ longlong2 stats_sum(__amdgpu_buffer_rsrc_t pr, int off, int zbytes, int cw, int mb_first)
{
    long long a = 0, b = 0;
    for (int mb0 = mb_first; mb0 < cw; mb0 += 16) {
        uint4 t[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            t[i] = make_uint4(0, 0, 0, 0);
            if (mb0 + i < cw) t[i] = ld16_sc1(pr, (mb0 + i) * zbytes + off);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const longlong2 v = as_ll2(t[i]);
            a += v.x;
            b += v.y;
        }
    }
    return make_longlong2(a, b);
}

// the epilogue of one 16 x 16 output tile (column tile n0, rows mrow0 ..) in two halves: the operands as k_gemm_skinny / k_gemm_dec
// fetch them (sc1 loads for what other workgroups of this launch have produced: partial row statistics, residual rows) ...
struct EpiOps {
    longlong2 st_a, st_r;
    float4 pc1, pbias, pgam, pbeta;
    uint2 presid;
    int em, n, edyn;
    bool exok;
};
struct StatsRef {
    __amdgpu_buffer_rsrc_t part;
    int zbytes, cw, aln_off, rln_off;
};
__device__ __forceinline__ EpiOps epi_load(const IgemmParams &p, int M, int n0, int mrow0, int fr, int fg, bool with_resid)
{
    EpiOps e;
    e.em = mrow0 + fr;
    e.exok = e.em < M;
    e.edyn = dyn_value(p, e.exok ? e.em : 0);
    e.n = n0 + 4 * fg;
    e.st_a = make_longlong2(0, 0);
    e.st_r = make_longlong2(0, 0);
    e.pc1 = make_float4(0.f, 0.f, 0.f, 0.f);
    e.pbias = e.pc1;
    e.pgam = e.pc1;
    e.pbeta = e.pc1;
    e.presid = make_uint2(0, 0);
    const bool ln_mode = p.aln_stats || p.rln_stats || p.stats_out;
    if (ln_mode) {
        if (e.n < p.N) {
            if (p.aln_stats && !p.ln_rms) e.pc1 = *reinterpret_cast<const float4 *>(p.aln_c1 + e.n);
            if (p.bias) e.pbias = *reinterpret_cast<const float4 *>(p.bias + e.n);
            if (p.resid && p.rln_stats) {
                e.pgam = *reinterpret_cast<const float4 *>(p.rln_gamma + e.n);
                e.pbeta = *reinterpret_cast<const float4 *>(p.rln_beta + e.n);
            }
            if (with_resid && p.resid && e.exok) e.presid = ld8_sc1(p.resid + epi_row(p, e.em, e.n, e.edyn).rbase + e.n);
        }
    } else if (with_resid && e.exok && e.n < p.N && p.resid) {
        e.presid = ld8_sc1(p.resid + epi_row(p, e.em, e.n, e.edyn).rbase + e.n);
    }
    return e;
}
// ... and the shared arithmetic on the accumulated 4-vector
__device__ __forceinline__ void epi_finish(const IgemmParams &p, const EpiOps &e, f32x4 s, int fg)
{
    const bool ln_mode = p.aln_stats || p.rln_stats || p.stats_out;
    if (ln_mode)
        ln_epi4(p, e.em, e.n, e.exok, s, e.edyn, ln_row(p, e.st_a, e.st_r), e.pc1, e.pbias, e.pgam, e.pbeta, e.presid, fg);
    else if (e.exok && e.n < p.N)
        (void)igemm_store4_fast<true>(p, e.em, e.n, s, e.edyn, e.presid);
}

// A GEMM phase of one workgroup: its share of the column tiles x the K chains = tasks; a wave walks its tasks one after the other
// (the next task's weights are requested before the current one's MFMAs), every task leaves its partial tile in LDS, ONE barrier,
// then every output tile's chains are added in chain order and finished by the shared epilogue.  K beyond the LDS image's capacity
// (fc2: 3072) is walked in two halves of whole chains.
constexpr int MAXTASK = 16;       // partial tiles kept in LDS per pass (more tasks: several passes)
// debug: where thread 0 of cluster 0 / member 0 spends a GEMM phase (100 MHz ticks summed over all GEMM phases into g_prof[200 + I])
#define GSTAMP(I)                                                            \
    if (gprof) {                                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();    \
        gprof[200 + (I)] += now_ - gt_;                                      \
        gt_ = now_;                                                          \
    }
template <int MT, typename WaitFn>
__device__ __forceinline__ void phase_gemm(const IgemmParams &p, int ksplit, int so_off, const StatsRef &sr, int r0, int member, uint16_t *As,
                                           f32x4 *red, longlong2 *s_ln, WaitFn wait, unsigned long long *gprof)
{
    unsigned long long gt_ = gprof ? __builtin_amdgcn_s_memrealtime() : 0;
    constexpr int RB = 16 * MT;
    constexpr int UB = IFH_STEP_UBL;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int M = p.nbatch * p.T_out;
    const int cw = sr.cw;
    const int ntile = (p.N + 15) / 16;
    const int t_lo = (int)((int64_t)member * ntile / cw), t_hi = (int)((int64_t)(member + 1) * ntile / cw);
    const int nk = (p.K + 31) / 32;
    const int per = (nk + ksplit - 1) / ksplit;
    const int nh = p.K > KCAP ? 2 : 1;                 // K halves (host: whole chains per half, each within the image)
    const int cph = ksplit / nh;                       // chains per half
    const int tpp = MAXTASK / ksplit;                  // tiles per pass
    const bool ln_mode = p.aln_stats || p.rln_stats || p.stats_out;
    // task i of (pass tile range starting at ta, half h): tile ta + i / cph, chain h cph + i % cph
    auto task_of = [&](int i, int ta, int h, int &ks0, int &ks1, const uint16_t *&wrow, bool &wok) __attribute__((always_inline)) {
        const int n0 = (ta + i / cph) * 16;
        const int c = h * cph + i % cph;
        ks0 = c * per;
        ks1 = min(nk, ks0 + per);
        wok = n0 + fr < p.N;
        wrow = p.w + (int64_t)(wok ? n0 + fr : 0) * p.K + fg * 8;
    };
    // A wave's tasks of one (pass, half): i = wid, wid + SW (MAXTASK / SW = 2 at most).  The first UB k-steps of ALL of
    // them are requested at once -- for the first (pass, half) ahead of the cluster wait: weights depend on nothing the cluster
    // produces -- so that a phase waits for memory once, not once per task.
    constexpr int TPW = MAXTASK / SW;
    uint4 wv[TPW][UB];
    int ks0[TPW], ks1[TPW];
    const uint16_t *wrow[TPW];
    bool wok[TPW];
    auto fetch = [&](int ta, int tb, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < TPW; q++) {
            ks0[q] = ks1[q] = 0;
            wrow[q] = p.w;
            wok[q] = false;
            const int i = wid + q * SW;
            if (i < (tb - ta) * cph) {
                task_of(i, ta, h, ks0[q], ks1[q], wrow[q], wok[q]);
                load_w<UB>(wrow[q], wok[q], p.K, fg, ks0[q], ks1[q], wv[q]);
            }
        }
    };
    fetch(t_lo, min(t_hi, t_lo + tpp), 0);
    // the activation image: thread -> (row, 16-byte column c16, c16 + 16, ...) of the 32-row block; the row's address is fixed
    const int arow = tid >> 4, ac = tid & 15;
    const RowAddr ara = x_row(p, r0 + arow, M);
    GSTAMP(0)      // table + first weights requested
    wait();
    GSTAMP(1)      // cluster wait
    const __amdgpu_buffer_rsrc_t xr = mk_rsrc(p.x);
    if (so_off >= 0 && tid < 2 * RB) g_epi_stats[tid >> 1][tid & 1] = 0ull;
    // the rows' LayerNorm statistics = the cw workgroups' partial sums: threads 0..RB-1 the GEMM operand's rows, RB..2RB-1 the
    // residual's; the first 16 partials are requested here, ahead of the activation image (one wait for both)
    uint4 st_t[16];
    int st_off = -1;
    if (ln_mode && tid < 2 * RB) {
        const int row = tid & (RB - 1), which = tid / RB;
        if (r0 + row < M && (which == 0 ? p.aln_stats != nullptr : p.rln_stats != nullptr))
            st_off = (which == 0 ? sr.aln_off : sr.rln_off) + (r0 + row) * 16;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        st_t[i] = make_uint4(0, 0, 0, 0);
        if (st_off >= 0 && i < cw) st_t[i] = ld16_sc1(sr.part, i * sr.zbytes + st_off);
    }
    // the first epilogue item's residual rows (produced by the cluster) travel with the activations
    uint2 r0pre = make_uint2(0, 0);
    if (p.resid && wid < (min(t_hi, t_lo + tpp) - t_lo) * MT) {
        const int em = r0 + (wid % MT) * 16 + fr, n = (t_lo + wid / MT) * 16 + 4 * fg;
        if (em < M && n < p.N) r0pre = ld8_sc1(p.resid + epi_row(p, em, n, dyn_value(p, em)).rbase + n);
    }
    for (int ta = t_lo; ta < t_hi; ta += tpp) {
        const int tb = min(t_hi, ta + tpp);
        for (int h = 0; h < nh; h++) {
            const bool first = h == 0 && ta == t_lo;
            if (!first) {
                __syncthreads();                       // the image / the partial tiles are free again
                fetch(ta, tb, h);
            }
            // the row block's activations for this half -> LDS (zeros for rows beyond M and for k beyond K)
            const int kbase = h * cph * per * 32;
            const int kw = nh == 1 ? nk * 32 : cph * per * 32;      // image width
            const int vpr = kw / 8;
            {
                static_assert(RB == 32 && SW * 64 == 16 * RB, "activation image: 16 threads per row");
                constexpr int AV = 6;                  // 16-byte vectors per thread per batch: 768 of K (every shape but ps and fc2)
                for (int q0 = 0; q0 * 16 < vpr; q0 += AV) {
                    uint4 v[AV];
#pragma unroll
                    for (int q = 0; q < AV; q++) {
                        const int kv = ac + 16 * (q0 + q);
                        v[q] = make_uint4(0, 0, 0, 0);
                        if (kv < vpr && ara.ok && kbase + kv * 8 < p.K) v[q] = ld16_sc1(xr, (ara.off + kbase + kv * 8) * 2);
                    }
#pragma unroll
                    for (int q = 0; q < AV; q++) {
                        const int kv = ac + 16 * (q0 + q);
                        if (kv < vpr) *reinterpret_cast<uint4 *>(&As[arow * LDK + kv * 8]) = v[q];
                    }
                }
            }
            // ... and the rows' statistics: the partial sums requested ahead of the image are added up
            if (first && ln_mode && tid < 2 * RB) {
                long long sa = 0, sb = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const longlong2 v = as_ll2(st_t[i]);
                    sa += v.x;
                    sb += v.y;
                }
                if (cw > 16 && st_off >= 0) {
                    const longlong2 v = stats_sum(sr.part, st_off, sr.zbytes, cw, 16);
                    sa += v.x;
                    sb += v.y;
                }
                s_ln[tid] = make_longlong2(sa, sb);
            }
            __syncthreads();
            GSTAMP(2)      // statistics + activation image
            const uint16_t *Ah = As - kbase;           // fragment address of absolute k
#pragma unroll
            for (int q = 0; q < TPW; q++) {
                const int i = wid + q * SW;
                if (i < (tb - ta) * cph) {
                    f32x4 acc[MT];
#pragma unroll
                    for (int j = 0; j < MT; j++) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    gemm_chain_lds<MT, UB>(Ah, wrow[q], wok[q], p.K, ks0[q], ks1[q], fr, fg, wv[q], acc);
                    const int slot = (i / cph) * ksplit + h * cph + i % cph;
#pragma unroll
                    for (int j = 0; j < MT; j++) red[(slot * MT + j) * 64 + lane] = acc[j];
                }
            }
        }
        __syncthreads();
        GSTAMP(3)      // tasks
        // every tile of the pass: chains added in chain order, then the shared epilogue
        for (int it = wid; it < (tb - ta) * MT; it += SW) {
            const int tl = it / MT, j = it - tl * MT;
            f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < ksplit; c++) sacc += red[((tl * ksplit + c) * MT + j) * 64 + lane];
            const bool pre = it == wid && ta == t_lo;
            EpiOps e = epi_load(p, M, (ta + tl) * 16, r0 + j * 16, fr, fg, !pre);
            if (pre) e.presid = r0pre;
            e.st_a = s_ln[j * 16 + fr];
            e.st_r = s_ln[RB + j * 16 + fr];
            epi_finish(p, e, sacc, fg);
        }
    }
    GSTAMP(4)      // epilogue
    // this workgroup's partial row statistics of the phase's output rows (zeros when it had no tile: the consumer adds all cw)
    if (so_off >= 0) {
        __syncthreads();
        if (tid < RB && r0 + tid < M) {
            const unsigned long long a = g_epi_stats[tid][0], b = g_epi_stats[tid][1];
            st16_out(sr.part, member * sr.zbytes + so_off + (r0 + tid) * 16,
                     make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)));
        }
    }
}

// ---- attention phase -----------------------------------------------------------------------------------------------------------
// One wave per (row, head).  NW = 4 reproduces a four-wave k_attn_decode workgroup: virtual wave w walks keys w*8 + g + 128 i + 32 u
// with its own online-softmax state; the four states are merged over the 8 key groups and then in wave order, as the kernel's LDS
// exchange does.  All 32 (NW = 4) / 16 (NW = 1, two of its iterations) K|V loads of a 128- / 64-key span are in flight at once.
template <int NW>
__device__ __forceinline__ void attn_unit(const AttnArgs &P, int row, int h, int lane)
{
    constexpr int KU = 4;
    constexpr int NV = NW == 4 ? 4 : 2;                  // state sets processed per load batch: 4 virtual waves, or 2 iterations of one
    constexpr int SPAN = NW == 4 ? 128 : 64;             // keys per load batch
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
    const int kvoff = (h * 64 + 8 * c) * 2, kv_ts2 = (int)(P.kv_ts * 2);
    float m[NW], l[NW];
    f32x2 o2[NW][4];
#pragma unroll
    for (int w = 0; w < NW; w++) {
        m[w] = -1e30f;
        l[w] = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) o2[w][e] = (f32x2){0.0f, 0.0f};
    }
    for (int base = 0; base < klen; base += SPAN) {
        // two state sets per load batch (16 K|V loads in flight per lane): virtual waves 2 hb, 2 hb + 1 of the four, or the two
        // iterations of the one
#pragma unroll
        for (int hb = 0; hb < NV / 2; hb++) {
            uint4 kk[2][KU], vv[2][KU];
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const int s = 2 * hb + s2;
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    // NW = 4: virtual wave s, its key u of this iteration; NW = 1: iteration s of the one wave
                    const int key = NW == 4 ? base + s * 8 + g + u * 32 : base + s * 32 + g + u * 8;
                    kk[s2][u] = make_uint4(0, 0, 0, 0);
                    vv[s2][u] = make_uint4(0, 0, 0, 0);
                    if (key < klen) {
                        // (self-attention rows are read once per step: non-temporal; the cross-attention K|V measured twice as fast without)
                        kk[s2][u] = NW == 4 ? ld16_sc1_nt(kr, kvoff + key * kv_ts2) : ld16_sc1(kr, kvoff + key * kv_ts2);
                        vv[s2][u] = NW == 4 ? ld16_sc1_nt(vr, kvoff + key * kv_ts2) : ld16_sc1(vr, kvoff + key * kv_ts2);
                    }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const int s = 2 * hb + s2;
                const int key0 = NW == 4 ? base + s * 8 + g : base + s * 32 + g;
                if (key0 < klen) {                            // (the kernel's loop condition for this lane's key group)
                    bool valid[KU];
#pragma unroll
                    for (int u = 0; u < KU; u++) valid[u] = (NW == 4 ? key0 + u * 32 : key0 + u * 8) < klen;
                    f32x2 klo[KU / 2][4], khi[KU / 2][4], vp[KU][4];
                    attn_unpack<KU>(kk[s2], vv[s2], klo, khi, vp);
                    constexpr int W0 = 0;
                    attn_row_update<KU>(q2, klo, khi, vp, valid, m[NW == 4 ? s : W0], l[NW == 4 ? s : W0], o2[NW == 4 ? s : W0]);
                }
            }
        }
    }
    float mm = 0.f, ll = 0.f, oo[8];
#pragma unroll
    for (int w = 0; w < NW; w++) {
        float mw = m[w], lw = l[w], o[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            o[2 * e] = o2[w][e].x;
            o[2 * e + 1] = o2[w][e].y;
        }
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {            // the wave's 8 key groups
            const float m2 = __shfl_xor(mw, off, 64), l2 = __shfl_xor(lw, off, 64);
            const float mn = fmaxf(mw, m2);
            const float a = __expf(mw - mn), a2 = __expf(m2 - mn);
            lw = lw * a + l2 * a2;
#pragma unroll
            for (int i = 0; i < 8; i++) o[i] = o[i] * a + __shfl_xor(o[i], off, 64) * a2;
            mw = mn;
        }
        if (w == 0) {
            mm = mw;
            ll = lw;
#pragma unroll
            for (int i = 0; i < 8; i++) oo[i] = o[i];
        } else {                                             // the waves, in wave order
            const float mn = fmaxf(mm, mw);
            const float a = __expf(mm - mn), a2 = __expf(mw - mn);
            ll = ll * a + lw * a2;
#pragma unroll
            for (int i = 0; i < 8; i++) oo[i] = oo[i] * a + o[i] * a2;
            mm = mn;
        }
    }
    if (g == 0) {
        const float inv = ll > 0.0f ? 1.0f / ll : 0.0f;
        uint4 pk;
        pk.x = pack2(oo[0] * inv, oo[1] * inv);
        pk.y = pack2(oo[2] * inv, oo[3] * inv);
        pk.z = pack2(oo[4] * inv, oo[5] * inv);
        pk.w = pack2(oo[6] * inv, oo[7] * inv);
        st16_out(mk_rsrc(P.ao), (int)(((int64_t)row * P.o_bs + h * 64 + 8 * c) * 2), pk);
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

// ---- stop rule and position advance (k_tts_stop_advance_rows; the statistics it clears are untouched by this kernel) ----
__device__ __forceinline__ void phase_stop(const StopArgs &P, int r0, int rows_here, int member)
{
    if (member != 0) return;
    const int tid = threadIdx.x;
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
                                                           unsigned long long *__restrict__ flags, unsigned long long *__restrict__ epochs,
                                                           int *__restrict__ xccw, unsigned char *__restrict__ part, int zbytes,
                                                           int *__restrict__ err, int *__restrict__ dbg_xcc,
                                                           unsigned long long *__restrict__ prof, int force_wt)
{
    constexpr int RB = 16 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint16_t *As = reinterpret_cast<uint16_t *>(lds_raw);
    f32x4 *red = reinterpret_cast<f32x4 *>(lds_raw + RB * LDK * 2);
    longlong2 *s_ln = reinterpret_cast<longlong2 *>(lds_raw + RB * LDK * 2 + MAXTASK * MT * 64 * sizeof(f32x4));
    const int b = blockIdx.x;
    const int j = b >> 3, kc = j / cw, member = j - kc * cw;
    const int cluster = (b & 7) + 8 * kc;
    const int my_xcc = __builtin_amdgcn_s_getreg(63508);      // HW_REG_XCC_ID
    if (dbg_xcc && threadIdx.x == 0) dbg_xcc[b] = my_xcc;
    const int r0 = cluster * RB;
    if (r0 >= nrows) return;
    const int rows_here = min(RB, nrows - r0);
    unsigned long long *cflags = flags + (int64_t)cluster * CWMAX * LINE;
    unsigned long long *my_flag = cflags + (int64_t)member * LINE;
    int *cxcc = xccw + (int64_t)cluster * CWMAX;
    if (threadIdx.x == 0) {
        g_epi_plain = 0;                                       // write-through until the cluster's placement is known
        __hip_atomic_store(cxcc + member, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the flag value when this launch began: left by the previous launch (member 0 stores it once every member has read it, i.e.
    // after the first wait)
    const unsigned long long e0 = __hip_atomic_load(epochs + (int64_t)cluster * LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    StatsRef sr;
    sr.part = mk_rsrc(part);
    sr.zbytes = zbytes;
    sr.cw = cw;
    const PhaseTab ctab = (PhaseTab)tab;
    // prof (debug): 100 MHz ticks cluster 0 / member 0 spent per phase -- [2 ph] waiting for the cluster, [2 ph + 1] in the phase
    const bool stamp = prof && cluster == 0 && member == 0 && threadIdx.x == 0;
    unsigned long long tprev = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
    for (int ph = 0; ph < nphase; ph++) {
        const int kind = ctab[ph].kind;
        auto wait = [&]() __attribute__((always_inline)) {
            if (ph == 0) return;
            cluster_wait(cflags, cw, e0 + (unsigned long long)ph, err);
            if (ph == 1) {
                // every member has published its XCC id: one L2 for the whole cluster -> plain stores from here on
                if (threadIdx.x < 64) {
                    const int lane = threadIdx.x;
                    const int x = __hip_atomic_load(cxcc + min(lane, cw - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool same = !__any(x != my_xcc);
                    if (lane == 0) {
                        g_epi_plain = (same && !force_wt) ? 1 : 0;
                        if (member == 0) epochs[(int64_t)cluster * LINE] = e0 + (unsigned long long)(nphase - 1);
                    }
                }
                __syncthreads();
            }
            if (stamp) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                prof[2 * ph] += now - tprev;
                tprev = now;
            }
        };
        if (kind == PH_GEMM) {
            const IgemmParams p = ld_const(&ctab[ph].g);
            sr.aln_off = ctab[ph].aln_off;
            sr.rln_off = ctab[ph].rln_off;
            phase_gemm<MT>(p, ctab[ph].ksplit, ctab[ph].so_off, sr, r0, member, As, red, s_ln, wait, stamp ? prof : (unsigned long long *)nullptr);
            if (stamp) tprev = __builtin_amdgcn_s_memrealtime(), prof[206] += 1;
        } else if (kind == PH_ATTN) {
            const AttnArgs a = ld_const(&ctab[ph].a);
            wait();
            phase_attn(a, r0, rows_here, member, cw);
        } else {
            const StopArgs sa = ld_const(&ctab[ph].s);
            wait();
            phase_stop(sa, r0, rows_here, member);
        }
        if (ph + 1 < nphase) cluster_arrive(my_flag, e0 + (unsigned long long)ph + 1);
        if (stamp) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            prof[2 * ph + 1] += now - tprev;
            if (kind == PH_GEMM) prof[205] += now - tprev;      // statistics partials + arrive
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
    int64_t zbytes = 0;                       // size of the statistics array the step's LayerNorm phases address
};
struct StepCtx {
    unsigned long long *flags = nullptr, *epochs = nullptr;      // [clusters][CWMAX] / [clusters] words, each on its own line
    int *xccw = nullptr;                      // [clusters][CWMAX]: the XCC id every workgroup of the running launch reported
    unsigned char *part = nullptr;            // [CWMAX][zbytes]: the workgroups' partial row statistics
    int64_t part_zbytes = 0;
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
    {
        // the LDS image holds KCAP of K: deeper K is walked as two halves of whole chains
        const int ks = p.K >= 2048 ? 4 : 2, nk = (p.K + 31) / 32, per = (nk + ks - 1) / ks;
        if (p.K > KCAP && !(ks == 4 && 2 * per * 32 <= KCAP && 4 * per == nk))
            return fail(IFH_EINVAL, "step record: K beyond the resident kernel's activation image");
    }
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
    if ((((uintptr_t)prob_logits) & 7) || logits_ld % 2 || zero_bytes >= (1ll << 24))
        return fail(IFH_EINVAL, "step record: stop rule layout");
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
    s.s.zbytes = zero_bytes;
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
    // the LayerNorm phases address the statistics array the stop rule clears: their offsets into it select the partial sums
    const unsigned char *zb = nullptr;
    int64_t zbytes = 0;
    for (const StepPhase &ph : g_rec.ph)
        if (ph.kind == PH_STOP) {
            zb = (const unsigned char *)ph.s.zbuf;
            zbytes = ph.s.zbytes;
        }
    for (StepPhase &ph : g_rec.ph) {
        ph.so_off = ph.aln_off = ph.rln_off = -1;
        if (ph.kind != PH_GEMM) continue;
        const void *ptrs[3] = {ph.g.stats_out, ph.g.aln_stats, ph.g.rln_stats};
        int *offs[3] = {&ph.so_off, &ph.aln_off, &ph.rln_off};
        for (int i = 0; i < 3; i++) {
            if (!ptrs[i]) continue;
            const int64_t off = (const unsigned char *)ptrs[i] - zb;
            if (!zb || off < 0 || off % 16 != 0 || off + (int64_t)g_rec.rows * 16 > zbytes) {
                g_rec.ph.clear();
                return fail(IFH_EINVAL, "step record: LayerNorm statistics outside the array the stop rule clears");
            }
            *offs[i] = (int)off;
        }
    }
    StepProg *pr = new StepProg;
    pr->nphase = (int)g_rec.ph.size();
    pr->nrows = g_rec.rows;
    pr->zbytes = zbytes;
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
    c->max_clusters = (max_rows + 31) / 32 + 8;
    const size_t nflag = sizeof(unsigned long long) * LINE * CWMAX * c->max_clusters, nep = sizeof(unsigned long long) * LINE * c->max_clusters;
    const size_t nxcc = sizeof(int) * CWMAX * c->max_clusters;
    hipError_t e = hipMalloc((void **)&c->flags, nflag);
    if (e == hipSuccess) e = hipMalloc((void **)&c->epochs, nep);
    if (e == hipSuccess) e = hipMalloc((void **)&c->xccw, nxcc);
    if (e == hipSuccess) e = hipMalloc((void **)&c->err, 256);
    if (e == hipSuccess) e = hipMalloc((void **)&c->dbg, sizeof(int) * 4096);
    if (e == hipSuccess) e = hipMalloc((void **)&c->prof, sizeof(unsigned long long) * 256);
    if (e == hipSuccess) e = hipMemset(c->flags, 0, nflag);
    if (e == hipSuccess) e = hipMemset(c->epochs, 0, nep);
    if (e == hipSuccess) e = hipMemset(c->xccw, 0xff, nxcc);
    if (e == hipSuccess) e = hipMemset(c->err, 0, 256);
    if (e == hipSuccess) e = hipMemset(c->dbg, 0xff, sizeof(int) * 4096);
    if (e == hipSuccess) e = hipMemset(c->prof, 0, sizeof(unsigned long long) * 256);
    if (e != hipSuccess) {
        (void)ifh_step_ctx_destroy(c);
        return check_hip(e, "step ctx alloc");
    }
    *ctx_out = c;
    return IFH_OK;
}

extern "C" int ifh_step_ctx_destroy(ifh_step_ctx_t ctx)
{
    StepCtx *c = reinterpret_cast<StepCtx *>(ctx);
    if (!c) return IFH_OK;
    (void)hipFree(c->flags);
    (void)hipFree(c->epochs);
    (void)hipFree(c->xccw);
    (void)hipFree(c->part);
    (void)hipFree(c->err);
    (void)hipFree(c->dbg);
    (void)hipFree(c->prof);
    delete c;
    return IFH_OK;
}

// the error word (1: a cluster wait ran into its 2 s bound) and, when xcc_out is given, the XCC id every workgroup of the last
// launch with debug bit 0 ran on (n_xcc ints, -1 where no workgroup wrote).  Synchronises the device.  An error clears the flags.
extern "C" int ifh_step_ctx_status(ifh_step_ctx_t ctx, int *err_out, int *xcc_out, int n_xcc)
{
    StepCtx *c = reinterpret_cast<StepCtx *>(ctx);
    IFH_CHECK_ARG(c && err_out && n_xcc >= 0 && n_xcc <= 4096);
    if (int rc = check_hip(hipDeviceSynchronize(), "step ctx sync")) return rc;
    if (int rc = check_hip(hipMemcpy(err_out, c->err, sizeof(int), hipMemcpyDeviceToHost), "step ctx err")) return rc;
    if (xcc_out && n_xcc)
        if (int rc = check_hip(hipMemcpy(xcc_out, c->dbg, sizeof(int) * n_xcc, hipMemcpyDeviceToHost), "step ctx xcc")) return rc;
    if (*err_out) {
        (void)hipMemset(c->flags, 0, sizeof(unsigned long long) * LINE * CWMAX * c->max_clusters);
        (void)hipMemset(c->epochs, 0, sizeof(unsigned long long) * LINE * c->max_clusters);
        (void)hipMemset(c->err, 0, 256);
    }
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

extern "C" int ifh_step_run(ifh_step_prog_t prog, ifh_step_ctx_t ctx, int cw, int debug, ifh_stream_t stream)
{
    StepProg *pr = reinterpret_cast<StepProg *>(prog);
    StepCtx *c = reinterpret_cast<StepCtx *>(ctx);
    IFH_CHECK_ARG(pr && c && cw >= 1 && cw <= CWMAX && pr->nphase <= 128);
    constexpr int MT = 2, RB = 16 * MT;
    const int nclusters = (pr->nrows + RB - 1) / RB;
    const int groups = (nclusters + 7) / 8;
    IFH_CHECK_ARG(groups * 8 <= c->max_clusters && groups * 8 * cw <= 4096);
    if (pr->zbytes > 0 && c->part_zbytes != pr->zbytes) {
        // the workgroups' partial statistics: one image of the statistics array per cluster member (sized on first use; a context
        // serves one decode state, so this happens once)
        if (c->part) {
            if (int rc = check_hip(hipDeviceSynchronize(), "step ctx sync")) return rc;
            (void)hipFree(c->part);
            c->part = nullptr;
        }
        if (int rc = check_hip(hipMalloc((void **)&c->part, (size_t)CWMAX * pr->zbytes), "step ctx partials")) return rc;
        c->part_zbytes = pr->zbytes;
    }
    constexpr size_t lds = (size_t)RB * LDK * 2 + (size_t)MAXTASK * MT * 64 * sizeof(f32x4) + 2 * RB * sizeof(longlong2);
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_step_resident<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return check_hip(e, "step_resident lds attr");
        attr_once.done(attr_dev);
    }
    hipLaunchKernelGGL(k_step_resident<MT>, dim3(groups * 8 * cw), dim3(SW * 64), lds, as_stream(stream), (const StepPhase *)pr->tab,
                       pr->nphase, pr->nrows, cw, c->flags, c->epochs, c->xccw, c->part, (int)pr->zbytes, c->err,
                       (debug & 1) ? c->dbg : (int *)nullptr, (debug & 2) ? c->prof : (unsigned long long *)nullptr, (debug & 4) ? 1 : 0);
    IFH_LAUNCH_CHECK("step_resident");
    return IFH_OK;
}
