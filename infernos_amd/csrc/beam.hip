// Beam search over the per-token decode step, at the width and length penalty CTranslate2's Whisper.generate uses by default
// for InfernSTTWorker.infer_and_decode_ct2 (Cluster/InfernSTTWorker.py:61-75; beam_size 5, length_penalty 1).  The
// bookkeeping is transformers' GenerationMixin._beam_search -- NOT CTranslate2's own termination rule, which is absent from
// the image and unpinned -- the formulation the parity fixtures pin (include/infernos_hip.h: ifh_beam_desc): 2K candidates per batch
// item per step, the best K that did not just end keep running, candidates ranked inside the first K that end
// (eos or length) compete with the K finished hypotheses kept so far on sum(log p) / length ** length_penalty, and
// a batch item stops improving once its best running beam cannot beat its worst finished one.
//
// Everything lives on the device and is indexed through the step counter the decode graph advances, so a step
// (decoder launches + ifh_beam_step + ifh_kv_gather_bf16 per layer) is one hipGraph replay with no host round trip:
//   k_beam_rowtop  one workgroup per decode row (batch item x beam): log-sum-exp of the row's logits and its 16 (sampling:
//                  32) best masked logits -- thread maxima give a threshold, a second pass lists what reaches it, 16-byte loads.
//   k_beam_update  one wave per batch item: K-way merge of the rows' lists into the 2K candidates, then the running /
//                  finished bookkeeping, the permutation of the token matrix columns and the source row of every
//                  running beam for the KV gather.
//   k_kv_gather    dst[row] = src[beam_src[row]] for the first cur_len tokens of a self-attention KV cache.
// The row stage also serves token sampling for the LLM worker (k_rep_penalty, k_sample_pick further down).
#include "common.h"

namespace ifh {

constexpr int BEAM_MAXK = 8;       // beams
constexpr int BEAM_SLOTS = 16;     // candidates kept per row (>= 2 * beams)
constexpr float BEAM_NEG = -1.0e9f;

__device__ __forceinline__ bool beam_better(float v, int i, float ov, int oi) { return v > ov || (v == ov && i < oi); }

constexpr int BEAM_LIST = 1024;    // candidates a row may pass to the selection before the exhaustive path takes over

// block-wide best (value, then lower index) of one candidate per thread; every thread gets the result
__device__ __forceinline__ void beam_block_best(float &bv, int &bi, float *s_v, int *s_i)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (beam_better(ov, oi, bv, bi)) {
            bv = ov;
            bi = oi;
        }
    }
    __syncthreads();
    if (lane == 0) {
        s_v[wid] = bv;
        s_i[wid] = bi;
    }
    __syncthreads();
    bv = s_v[0];
    bi = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; w++)
        if (beam_better(s_v[w], s_i[w], bv, bi)) {
            bv = s_v[w];
            bi = s_i[w];
        }
}

// Per decode row: log-sum-exp of the raw logits and the BEAM_SLOTS best masked logits (value desc, token asc).
//   pass 1: every thread's log-sum-exp share and its single best candidate; the 16th best of the 256 thread maxima
//           is a threshold at least 16 candidates reach;
//   pass 2: the row again (L2-resident: the head GEMM just wrote it), candidates >= threshold appended to an LDS list
//           (a few dozen entries);
//   then 16 rounds of block arg-max over the list.  A row with more than BEAM_LIST candidates at the threshold (ties
//   en masse) takes 16 exhaustive scans instead: always exact.
// Keeping a sorted top-16 per thread in registers costs 5x more: with 64 lanes a wave takes the insertion branch on
// practically every element.
template <int SLOTS>
__global__ __launch_bounds__(256) void k_beam_rowtop(const float *__restrict__ logits, int64_t ld, int V,
                                                     const float *__restrict__ suppress,
                                                     const float *__restrict__ begin_suppress,
                                                     const int32_t *__restrict__ pos, int prompt_len,
                                                     float *__restrict__ cand_val, int32_t *__restrict__ cand_tok,
                                                     float *__restrict__ row_lse)
{
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    __shared__ float s_m[4], s_s[4];
    __shared__ float l_v[BEAM_LIST];
    __shared__ int l_i[BEAM_LIST];
    __shared__ int l_n;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float *x = logits + (int64_t)row * ld;
    const float *bs = (begin_suppress && pos[0] == prompt_len) ? begin_suppress : nullptr;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | (uintptr_t)(ld * 4) | reinterpret_cast<uintptr_t>(suppress) |
                       reinterpret_cast<uintptr_t>(begin_suppress)) & 15) == 0;
    const int V4 = vec ? (V >> 2) : 0;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    if (tid == 0) l_n = 0;
    // ---- pass 1
    float m = -INFINITY, s = 0.0f, best = -INFINITY;
    int besti = 0x7fffffff;
    // (SV / BV: the suppress / begin-suppress addends of the element, already loaded)
#define BEAM_P1(LV, IDX, SV, BV)                                  \
    {                                                             \
        const float lv_ = (LV);                                   \
        const int i_ = (IDX);                                     \
        if (lv_ > m) {                                            \
            s = s * __expf(m - lv_) + 1.0f;                       \
            m = lv_;                                              \
        } else {                                                  \
            s += __expf(lv_ - m);                                 \
        }                                                         \
        float c_ = lv_;                                           \
        if (suppress) c_ += (SV);                                 \
        if (bs) c_ += (BV);                                       \
        if (c_ > best) {                                          \
            best = c_;                                            \
            besti = i_;                                           \
        }                                                         \
    }
    // a thread's four vectors of logits and of both tables go out together, from clamped addresses (as `j < V4 ? x4[j] : 0` and
    // `if (suppress) c += suppress[i]` every load sat in its own exec branch with an s_waitcnt vmcnt(0) behind it: ~20 serial round
    // trips per 4 096 logits); what a thread adds, and in which order, is unchanged
    const float4 *sp4 = reinterpret_cast<const float4 *>(suppress), *bs4 = reinterpret_cast<const float4 *>(bs);
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#define BEAM_LOAD4()                                                                              \
    float4 q[4], sq[4], bq[4];                                                                    \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                               \
        const int j = base + tid + 256 * u;                                                       \
        const int jc = j < V4 ? j : V4 - 1;                                                       \
        q[u] = x4[jc];                                                                            \
        sq[u] = zero4;                                                                            \
        bq[u] = zero4;                                                                            \
        if (suppress) sq[u] = sp4[jc];                                                            \
        if (bs) bq[u] = bs4[jc];                                                                  \
    }
    for (int base = 0; base < V4; base += 256 * 4) {
        BEAM_LOAD4()
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = base + tid + 256 * u;
            if (j < V4) {
                BEAM_P1(q[u].x, 4 * j, sq[u].x, bq[u].x) BEAM_P1(q[u].y, 4 * j + 1, sq[u].y, bq[u].y)
                BEAM_P1(q[u].z, 4 * j + 2, sq[u].z, bq[u].z) BEAM_P1(q[u].w, 4 * j + 3, sq[u].w, bq[u].w)
            }
        }
    }
    for (int i = 4 * V4 + tid; i < V; i += 256) BEAM_P1(x[i], i, suppress[i], bs[i])
#undef BEAM_P1
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        const float mn = fmaxf(m, m2);
        s = (m == -INFINITY ? 0.0f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.0f : s2 * __expf(m2 - mn));
        m = mn;
    }
    if (lane == 0) {
        s_m[wid] = m;
        s_s[wid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        float mn = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float st = 0.0f;
        for (int w = 0; w < 4; w++)
            if (s_m[w] != -INFINITY) st += s_s[w] * __expf(s_m[w] - mn);
        row_lse[row] = mn + __logf(st);
    }
    // threshold: the SLOTS-th best thread maximum (-inf if fewer threads hold a finite candidate)
    float thr = -INFINITY;
    for (int r = 0; r < SLOTS; r++) {
        float bv = best;
        int bi = besti;
        beam_block_best(bv, bi, s_v, s_i);
        thr = bv;
        if (besti == bi && bi != 0x7fffffff) {
            best = -INFINITY;
            besti = 0x7fffffff;
        }
    }
    // ---- pass 2: candidates at or above the threshold (finite ones only)
#define BEAM_P2(LV, IDX, SV, BV)                                  \
    {                                                             \
        const int i_ = (IDX);                                     \
        float c_ = (LV);                                          \
        if (suppress) c_ += (SV);                                 \
        if (bs) c_ += (BV);                                       \
        if (c_ >= thr && c_ > -INFINITY) {                        \
            const int at = atomicAdd(&l_n, 1);                    \
            if (at < BEAM_LIST) {                                 \
                l_v[at] = c_;                                     \
                l_i[at] = i_;                                     \
            }                                                     \
        }                                                         \
    }
    for (int base = 0; base < V4; base += 256 * 4) {
        BEAM_LOAD4()
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = base + tid + 256 * u;
            if (j < V4) {
                BEAM_P2(q[u].x, 4 * j, sq[u].x, bq[u].x) BEAM_P2(q[u].y, 4 * j + 1, sq[u].y, bq[u].y)
                BEAM_P2(q[u].z, 4 * j + 2, sq[u].z, bq[u].z) BEAM_P2(q[u].w, 4 * j + 3, sq[u].w, bq[u].w)
            }
        }
    }
    for (int i = 4 * V4 + tid; i < V; i += 256) BEAM_P2(x[i], i, suppress[i], bs[i])
#undef BEAM_P2
#undef BEAM_LOAD4
    __syncthreads();
    const int n = l_n;
    if (n <= BEAM_LIST) {
        // thread t owns list entries t, t + 256, ...
        for (int r = 0; r < SLOTS; r++) {
            float bv = -INFINITY;
            int bi = 0x7fffffff, at = -1;
            for (int e = tid; e < n; e += 256)
                if (beam_better(l_v[e], l_i[e], bv, bi)) {
                    bv = l_v[e];
                    bi = l_i[e];
                    at = e;
                }
            const int mine = bi;
            beam_block_best(bv, bi, s_v, s_i);
            if (at >= 0 && mine == bi) {            // the winner's owner retires the entry
                l_v[at] = -INFINITY;
                l_i[at] = 0x7fffffff;
            }
            if (tid == 0) {
                cand_val[row * SLOTS + r] = bv;
                cand_tok[row * SLOTS + r] = bi == 0x7fffffff ? 0 : bi;
            }
        }
    } else {
        // exhaustive: round r scans the row for the best candidate after round r-1's pick in (value desc, token asc) order
        float pv = INFINITY;
        int pi = -1;
        for (int r = 0; r < SLOTS; r++) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int i = tid; i < V; i += 256) {
                float c_ = x[i];
                if (suppress) c_ += suppress[i];
                if (bs) c_ += bs[i];
                const bool after = c_ < pv || (c_ == pv && i > pi);
                if (after && c_ > -INFINITY && beam_better(c_, i, bv, bi)) {
                    bv = c_;
                    bi = i;
                }
            }
            beam_block_best(bv, bi, s_v, s_i);
            pv = bv;
            pi = bi;
            if (tid == 0) {
                cand_val[row * SLOTS + r] = bv;
                cand_tok[row * SLOTS + r] = bi == 0x7fffffff ? 0 : bi;
            }
            if (bi == 0x7fffffff) pv = -INFINITY;       // exhausted: the remaining slots stay (-inf, 0)
        }
    }
}


// ---- sampling (transformers' generate with do_sample: RepetitionPenaltyLogitsProcessor, then TemperatureLogitsWarper,
// TopKLogitsWarper, TopPLogitsWarper, softmax, one draw; generation/logits_process.py) ----
constexpr int SAMPLE_SLOTS = 32;       // top_k <= 32
constexpr int REP_VOCAB_MAX = 262144;   // one presence bit per vocabulary entry in LDS (32 KB)

// logits[row, t] = l < 0 ? l * penalty : l / penalty for every token t of the row's history, each token once however
// often it occurs and however long the history is: a presence bitmap over the vocabulary lives in LDS, and the thread
// whose atomicOr sets a token's bit is the only one that rewrites that logit.
__global__ __launch_bounds__(256) void k_rep_penalty(float *__restrict__ logits, int64_t ld, int V,
                                                     const int32_t *__restrict__ hist, int64_t hist_ld,
                                                     const int32_t *__restrict__ lens, float penalty)
{
    __shared__ uint32_t seen[REP_VOCAB_MAX / 32];
    const int row = blockIdx.x;
    float *x = logits + (int64_t)row * ld;
    const int32_t *h = hist + (int64_t)row * hist_ld;
    const int n = lens[row];
    const int words = (V + 31) / 32;
    for (int i = threadIdx.x; i < words; i += 256) seen[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int t = h[i];
        if (t < 0 || t >= V) continue;
        const uint32_t bit = 1u << (t & 31);
        if (atomicOr(&seen[t >> 5], bit) & bit) continue;       // another occurrence already owns this token
        const float v = x[t];
        x[t] = v < 0.0f ? v * penalty : v / penalty;
    }
}

// One thread per row over the row's SAMPLE_SLOTS best logits (value descending): temperature, top-k, top-p (a candidate
// stays iff the probability mass of it and everything below it exceeds 1 - top_p, the ascending cumulative sum of
// TopPLogitsWarper; the best one always stays), renormalise, inverse-CDF draw with the caller's uniform number.
__global__ __launch_bounds__(64) void k_sample_pick(const float *__restrict__ cand_val, const int32_t *__restrict__ cand_tok,
                                                    int nrows, float temperature, int top_k, float top_p,
                                                    const float *__restrict__ uniform, int32_t *__restrict__ out_tok,
                                                    float *__restrict__ out_probs /* optional [nrows][SAMPLE_SLOTS] */)
{
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= nrows) return;
    const float *cv = cand_val + (int64_t)row * SAMPLE_SLOTS;
    const int32_t *ct = cand_tok + (int64_t)row * SAMPLE_SLOTS;
    int k = min(top_k > 0 ? top_k : SAMPLE_SLOTS, SAMPLE_SLOTS);
    while (k > 1 && !(cv[k - 1] > -INFINITY)) k--;
    float z[SAMPLE_SLOTS], pr[SAMPLE_SLOTS];
    const float z0 = cv[0] / temperature;
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < SAMPLE_SLOTS; i++) {
        z[i] = i < k ? cv[i] / temperature : -INFINITY;
        pr[i] = i < k ? __expf(z[i] - z0) : 0.0f;
        sum += pr[i];
    }
    // top-p on softmax over the top-k set: cumulative sum in ascending order
    int keep = k;
    if (top_p < 1.0f) {
        float cum = 0.0f;
        keep = 1;
        for (int i = k - 1; i >= 1; i--) {
            cum += pr[i] / sum;
            if (cum > 1.0f - top_p) {
                keep = i + 1;
                break;
            }
        }
    }
    float ks = 0.0f;
    for (int i = 0; i < keep; i++) ks += pr[i];
    const float u = uniform[row] * ks;
    float acc = 0.0f;
    int pick = keep - 1;
    for (int i = 0; i < keep; i++) {
        acc += pr[i];
        if (acc > u) {
            pick = i;
            break;
        }
    }
    out_tok[row] = ct[pick];
    if (out_probs)
        for (int i = 0; i < SAMPLE_SLOTS; i++) out_probs[(int64_t)row * SAMPLE_SLOTS + i] = i < keep ? pr[i] / ks : 0.0f;
}

struct BeamParams {
    const float *cand_val;
    const int32_t *cand_tok;
    const float *row_lse;
    int32_t *toks;
    const int32_t *pos;
    float *run_scores, *fin_scores;
    int32_t *fin_seqs, *fin_len;
    uint8_t *is_fin;
    int32_t *unsat, *beam_src, *alive;
    int K, prompt_len, max_length, eos_id, nrows;
    float length_penalty;
};

__global__ __launch_bounds__(64) void k_beam_update(BeamParams p)
{
    extern __shared__ int32_t dyn[];       // [K][max_length] old token columns | [K][max_new] old finished rows
    __shared__ float c_s[BEAM_MAXK * BEAM_SLOTS];
    __shared__ int c_t[BEAM_MAXK * BEAM_SLOTS];
    __shared__ float top_s[BEAM_SLOTS];
    __shared__ int top_b[BEAM_SLOTS], top_t[BEAM_SLOTS], hit[BEAM_SLOTS];
    __shared__ int run_src[BEAM_MAXK], run_tok[BEAM_MAXK], fin_from[BEAM_MAXK], fin_len_old[BEAM_MAXK];
    // lane 0's work arrays: in LDS, not in (dynamically indexed, hence scratch-memory) private arrays
    __shared__ int head[BEAM_MAXK];
    __shared__ float new_run[BEAM_MAXK], ms[BEAM_MAXK + BEAM_SLOTS];
    __shared__ int mf[BEAM_MAXK + BEAM_SLOTS], used[BEAM_MAXK + BEAM_SLOTS];
    const int b = blockIdx.x, lane = threadIdx.x, K = p.K, KK = 2 * p.K;
    const int cur_len = p.pos[0], P = p.prompt_len, L = p.max_length, NN = p.max_length - p.prompt_len;
    if (cur_len >= L || cur_len < P) return;          // replays past the last position do nothing
    int32_t *seq_old = dyn, *fin_old = dyn + K * L;
    for (int i = lane; i < K * BEAM_SLOTS; i += 64) {
        const int k = i / BEAM_SLOTS, row = b * K + k;
        c_s[i] = (p.cand_val[row * BEAM_SLOTS + (i % BEAM_SLOTS)] - p.row_lse[row]) + p.run_scores[row];
        c_t[i] = p.cand_tok[row * BEAM_SLOTS + (i % BEAM_SLOTS)];
    }
    for (int i = lane; i < K * cur_len; i += 64) {
        const int j = i / K, k = i % K;
        seq_old[k * L + j] = p.toks[(int64_t)j * p.nrows + b * K + k];
    }
    for (int i = lane; i < K * NN; i += 64) fin_old[i] = p.fin_seqs[(int64_t)b * K * NN + i];
    __syncthreads();
    if (lane == 0) {
        // K-way merge of the per-row lists (each sorted, ties by token id): the 2K best continuations, ties by
        // (beam, token) like a top-k over the flattened [K * V] scores
        for (int k = 0; k < K; k++) head[k] = 0;
        bool all_hit = true;
        for (int r = 0; r < KK; r++) {
            int bk = 0;
            float bv = -INFINITY;
            bool any = false;
            for (int k = 0; k < K; k++) {
                if (head[k] >= BEAM_SLOTS) continue;
                const float v = c_s[k * BEAM_SLOTS + head[k]];
                if (!any || v > bv) {
                    bv = v;
                    bk = k;
                    any = true;
                }
            }
            top_s[r] = bv;
            top_b[r] = bk;
            top_t[r] = c_t[bk * BEAM_SLOTS + head[bk]];
            head[bk]++;
            hit[r] = (top_t[r] == p.eos_id) || (cur_len + 1 >= L);
            all_hit = all_hit && hit[r];
        }
        // running beams of the next step: the best K candidates that did not just end (then, if fewer, the rest)
        int n = 0;
        for (int pass = 0; pass < 2 && n < K; pass++)
            for (int r = 0; r < KK && n < K; r++)
                if ((hit[r] != 0) == (pass == 1)) {
                    run_src[n] = top_b[r];
                    run_tok[n] = top_t[r];
                    new_run[n] = top_s[r] + (hit[r] ? BEAM_NEG : 0.0f);
                    n++;
                }
        // finished hypotheses: the K kept so far against the candidates inside the first K that just ended
        const bool unsat = p.unsat[b] != 0;
        const float denom = __powf((float)(cur_len + 1 - P), p.length_penalty);
        for (int k = 0; k < K; k++) {
            ms[k] = p.fin_scores[b * K + k];
            mf[k] = p.is_fin[b * K + k] != 0;
            fin_len_old[k] = p.fin_len[b * K + k];
        }
        for (int r = 0; r < KK; r++) {
            const bool just = hit[r] && r < K;
            float v = top_s[r] / denom;
            if (!unsat) v += BEAM_NEG;
            if (!just) v += BEAM_NEG;
            ms[K + r] = v;
            mf[K + r] = just;
        }
        for (int i = 0; i < K + KK; i++) used[i] = 0;
        float worst = INFINITY;
        bool all_fin = true;
        for (int j = 0; j < K; j++) {
            int bi = -1;
            for (int i = 0; i < K + KK; i++)
                if (!used[i] && (bi < 0 || ms[i] > ms[bi])) bi = i;
            used[bi] = 1;
            fin_from[j] = bi;
            p.fin_scores[b * K + j] = ms[bi];
            p.is_fin[b * K + j] = mf[bi] ? 1 : 0;
            worst = fminf(worst, ms[bi]);
            all_fin = all_fin && mf[bi];
        }
        for (int k = 0; k < K; k++) p.run_scores[b * K + k] = new_run[k];
        // can the best running beam still beat the worst finished hypothesis?
        const float best_possible = new_run[0] / __powf((float)(cur_len + 1 - P), p.length_penalty);
        const bool still = unsat && (best_possible > (all_fin ? worst : BEAM_NEG));
        p.unsat[b] = still ? 1 : 0;
        if (still && !all_hit) p.alive[cur_len] = 1;
    }
    __syncthreads();
    // finished rows (generated part only) and their lengths
    for (int j = 0; j < K; j++) {
        const int src = fin_from[j];
        int32_t *dst = p.fin_seqs + ((int64_t)b * K + j) * NN;
        if (src < K) {
            if (src != j)
                for (int i = lane; i < NN; i += 64) dst[i] = fin_old[src * NN + i];
            if (lane == 0) p.fin_len[b * K + j] = fin_len_old[src];
        } else {
            const int r = src - K, col = top_b[r], n = cur_len - P;
            for (int i = lane; i < n; i += 64) dst[i] = seq_old[col * L + P + i];
            if (lane == 0) {
                dst[n] = top_t[r];
                p.fin_len[b * K + j] = n + 1;
            }
        }
    }
    // token matrix columns of the running beams, the token each feeds to the next step, and its source row
    for (int i = lane; i < K * cur_len; i += 64) {
        const int j = i / K, k = i % K;
        p.toks[(int64_t)j * p.nrows + b * K + k] = seq_old[run_src[k] * L + j];
    }
    if (lane < K) {
        p.toks[(int64_t)cur_len * p.nrows + b * K + lane] = run_tok[lane];
        p.beam_src[b * K + lane] = b * K + run_src[lane];
    }
}

__global__ __launch_bounds__(256) void k_kv_gather(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                   const int32_t *__restrict__ row_src,
                                                   const int32_t *__restrict__ len, int max_len, int64_t row_u4,
                                                   int tok_u4, int64_t layer_u4)
{
    const int row = blockIdx.y;
    const int n = min(len[0], max_len) * tok_u4;
    const uint4 *s = src + (int64_t)blockIdx.z * layer_u4 + (int64_t)row_src[row] * row_u4;
    uint4 *d = dst + (int64_t)blockIdx.z * layer_u4 + (int64_t)row * row_u4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}

}  // namespace ifh

using namespace ifh;

extern "C" int ifh_beam_step(const ifh_beam_desc *d, ifh_stream_t stream)
{
    IFH_CHECK_ARG(d && d->nbatch >= 0);
    if (d->nbatch == 0) return IFH_OK;
    IFH_CHECK_ARG(d->beams >= 1 && d->beams <= BEAM_MAXK && d->vocab >= 2 * d->beams && d->ld >= d->vocab);
    IFH_CHECK_ARG(d->logits && d->toks && d->pos && d->run_scores && d->fin_scores && d->fin_seqs && d->fin_len &&
                  d->is_fin && d->unsat && d->beam_src && d->alive && d->scratch);
    IFH_CHECK_ARG(d->prompt_len >= 1 && d->max_length > d->prompt_len && d->max_length <= 2048);
    IFH_CHECK_ARG(d->eos_id >= 0 && d->eos_id < d->vocab && d->length_penalty >= 0.0f);
    const int rows = d->nbatch * d->beams;
    float *cand_val = (float *)d->scratch;
    int32_t *cand_tok = (int32_t *)(cand_val + (size_t)rows * BEAM_SLOTS);
    float *row_lse = (float *)(cand_tok + (size_t)rows * BEAM_SLOTS);
    hipLaunchKernelGGL(k_beam_rowtop<BEAM_SLOTS>, dim3(rows), dim3(256), 0, as_stream(stream), d->logits, d->ld, d->vocab,
                       d->suppress, d->begin_suppress, d->pos, d->prompt_len, cand_val, cand_tok, row_lse);
    IFH_LAUNCH_CHECK("beam_rowtop");
    BeamParams p;
    p.cand_val = cand_val;
    p.cand_tok = cand_tok;
    p.row_lse = row_lse;
    p.toks = d->toks;
    p.pos = d->pos;
    p.run_scores = d->run_scores;
    p.fin_scores = d->fin_scores;
    p.fin_seqs = d->fin_seqs;
    p.fin_len = d->fin_len;
    p.is_fin = d->is_fin;
    p.unsat = d->unsat;
    p.beam_src = d->beam_src;
    p.alive = d->alive;
    p.K = d->beams;
    p.prompt_len = d->prompt_len;
    p.max_length = d->max_length;
    p.eos_id = d->eos_id;
    p.nrows = rows;
    p.length_penalty = d->length_penalty;
    const size_t dyn = (size_t)d->beams * (2 * d->max_length - d->prompt_len) * sizeof(int32_t);
    IFH_CHECK_ARG(dyn <= 60 * 1024);
    hipLaunchKernelGGL(k_beam_update, dim3(d->nbatch), dim3(64), dyn, as_stream(stream), p);
    IFH_LAUNCH_CHECK("beam_update");
    return IFH_OK;
}

extern "C" int ifh_kv_gather_bf16(const void *src, void *dst, const int32_t *row_src, const int32_t *len, int max_len,
                                  int nrows, int64_t row_stride, int tok_elems, int nlayers, int64_t layer_stride, ifh_stream_t stream)
{
    IFH_CHECK_ARG(nrows >= 0 && nlayers >= 0);
    if (nrows == 0 || nlayers == 0) return IFH_OK;
    IFH_CHECK_ARG(src && dst && src != dst && row_src && len && max_len >= 1 && nrows < 65536 && nlayers < 65536);
    IFH_CHECK_ARG(tok_elems > 0 && tok_elems % 8 == 0 && row_stride % 8 == 0 && row_stride >= (int64_t)max_len * tok_elems);
    IFH_CHECK_ARG(layer_stride % 8 == 0 && (nlayers == 1 || layer_stride >= (int64_t)nrows * row_stride));
    IFH_CHECK_ARG(((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0);
    hipLaunchKernelGGL(k_kv_gather, dim3(8, nrows, nlayers), dim3(256), 0, as_stream(stream), (const uint4 *)src, (uint4 *)dst,
                       row_src, len, max_len, row_stride / 8, tok_elems / 8, layer_stride / 8);
    IFH_LAUNCH_CHECK("kv_gather");
    return IFH_OK;
}

extern "C" int ifh_repetition_penalty_f32(float *logits, int64_t ld, int vocab, int nrows, const int32_t *history,
                                          int64_t hist_ld, const int32_t *lens, float penalty, ifh_stream_t stream)
{
    IFH_CHECK_ARG(nrows >= 0);
    if (nrows == 0 || penalty == 1.0f) return IFH_OK;
    IFH_CHECK_ARG(logits && history && lens && vocab > 0 && ld >= vocab && penalty > 0.0f && hist_ld > 0);
    IFH_CHECK_ARG(vocab <= REP_VOCAB_MAX);
    hipLaunchKernelGGL(k_rep_penalty, dim3(nrows), dim3(256), 0, as_stream(stream), logits, ld, vocab, history, hist_ld, lens,
                       penalty);
    IFH_LAUNCH_CHECK("rep_penalty");
    return IFH_OK;
}

extern "C" int ifh_sample_topk_f32(const float *logits, int64_t ld, int vocab, int nrows, float temperature, int top_k,
                                   float top_p, const float *uniform, int32_t *out_tokens, void *scratch,
                                   int32_t *out_cand /* optional [nrows][32] */, float *out_probs /* optional [nrows][32] */,
                                   ifh_stream_t stream)
{
    IFH_CHECK_ARG(nrows >= 0);
    if (nrows == 0) return IFH_OK;
    IFH_CHECK_ARG(logits && uniform && out_tokens && scratch && vocab > 0 && ld >= vocab);
    IFH_CHECK_ARG(temperature > 0.0f && top_k >= 0 && top_k <= SAMPLE_SLOTS && top_p > 0.0f && top_p <= 1.0f);
    float *cand_val = (float *)scratch;
    int32_t *cand_tok = (int32_t *)(cand_val + (size_t)nrows * SAMPLE_SLOTS);
    float *row_lse = (float *)(cand_tok + (size_t)nrows * SAMPLE_SLOTS);
    static const int32_t *no_pos = nullptr;
    hipLaunchKernelGGL(k_beam_rowtop<SAMPLE_SLOTS>, dim3(nrows), dim3(256), 0, as_stream(stream), logits, ld, vocab,
                       (const float *)nullptr, (const float *)nullptr, no_pos, 0, cand_val, cand_tok, row_lse);
    IFH_LAUNCH_CHECK("sample_rowtop");
    hipLaunchKernelGGL(k_sample_pick, dim3((nrows + 63) / 64), dim3(64), 0, as_stream(stream), cand_val, cand_tok, nrows,
                       temperature, top_k, top_p, uniform, out_tokens, out_probs);
    IFH_LAUNCH_CHECK("sample_pick");
    if (out_cand) {
        hipError_t e = hipMemcpyAsync(out_cand, cand_tok, (size_t)nrows * SAMPLE_SLOTS * sizeof(int32_t), hipMemcpyDeviceToDevice,
                                      as_stream(stream));
        if (e != hipSuccess) return check_hip(e, "sample_topk copy");
    }
    return IFH_OK;
}
