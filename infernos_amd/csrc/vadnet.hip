// vadnet.hip -- a recurrent speech-probability network for the per-window VAD step, shaped like the detector the reference loads.
//
// Core/VAD/SileroVAD.py:44-45 loads the third-party Silero VAD v3.1 TorchScript model (convolutional front end + a two-layer LSTM
// whose state, two [2, B, 64] tensors, the reference carries per channel: Core/VAD/SileroVADUtils.py:11,21-26,99,131).  The
// model file is not in the reference tree and cannot be fetched offline, so its WEIGHTS and exact layer list are not reproducible
// here (PARITY UNPINNED against Silero).  What CAN be reproduced is its interface and its cost class, so that the
// per-tick latency the benchmark reports contains a detector and not just an energy threshold:
//     x [768] -> Conv1d(1 -> 32, k 128, stride 64) + ReLU                      [11][32]
//             -> Conv1d(32 -> 64, k 3, stride 2, pad 1) + ReLU                 [6][64]
//             -> LSTM(64 -> 64) x 2 layers over the 6 steps, state (h, c) [2][64] carried from window to window
//             -> Linear(64 -> 1) per step -> sigmoid -> mean over the 6 steps   = speech probability
// (~1 MFLOP per window, the recurrent state of the reference's model, seeded weights: weights.synth_vadnet).  The arithmetic is
// pinned to the plain-PyTorch restatement the tests hold (itself checked against torch.nn.LSTM): tests/test_vadnet_gpu.py.
// One workgroup per call; weights are read transposed ([k][out]: consecutive lanes, consecutive outputs) from L2.
#include <math.h>

#include "common.h"

namespace ifh {

constexpr int VN_WIN = 768, VN_C1 = 32, VN_K1 = 128, VN_S1 = 64, VN_T1 = 11, VN_C2 = 64, VN_T2 = 6, VN_H = 64, VN_G = 4 * VN_H;
// weight blob (floats): w1t [128][32], b1 [32], w2t [3][32][64], b2 [64], per LSTM layer: wih_t [64][256], whh_t [64][256],
// bias [256] (b_ih + b_hh), then wout [64], bout [1]
constexpr int VN_OFF_W1 = 0, VN_OFF_B1 = VN_OFF_W1 + VN_K1 * VN_C1, VN_OFF_W2 = VN_OFF_B1 + VN_C1,
              VN_OFF_B2 = VN_OFF_W2 + 3 * VN_C1 * VN_C2, VN_OFF_L = VN_OFF_B2 + VN_C2, VN_LSTM = 2 * VN_H * VN_G + VN_G,
              VN_OFF_WO = VN_OFF_L + 2 * VN_LSTM, VN_OFF_BO = VN_OFF_WO + VN_H, VN_FLOATS = VN_OFF_BO + 1;

__device__ __forceinline__ float vn_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

// x row of call b: slot ? slot[b] : b (the block driver's window table is indexed by call slot); state and prob rows: b.
// h_out / c_out may be h_in / c_in: a workgroup reads its own rows before it writes them and touches no others.
__global__ __launch_bounds__(256) void k_vadnet(const float *x, const int32_t *__restrict__ slot, int n, const float *__restrict__ w,
                                                const float *h_in, const float *c_in, float *h_out, float *c_out,
                                                float *__restrict__ prob)
{
    __shared__ float xs[VN_WIN], f1[VN_T1][VN_C1], f2[VN_T2][VN_C2], hs[2][VN_H], cs[2][VN_H], gs[VN_G], ys[VN_T2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t xrow = slot ? slot[b] : b;
    for (int i = tid; i < VN_WIN; i += 256) xs[i] = x[xrow * VN_WIN + i];
    if (tid < 2 * VN_H) {
        const int l = tid / VN_H, j = tid % VN_H;
        hs[l][j] = h_in[((int64_t)l * n + b) * VN_H + j];
        cs[l][j] = c_in[((int64_t)l * n + b) * VN_H + j];
    }
    __syncthreads();
    // conv1: output (t, oc) = relu(b1[oc] + sum_k w1[oc][k] x[64 t + k]), k ascending
    for (int o = tid; o < VN_T1 * VN_C1; o += 256) {
        const int t = o / VN_C1, oc = o % VN_C1;
        float a = w[VN_OFF_B1 + oc];
        for (int k = 0; k < VN_K1; k++) a = fmaf(w[VN_OFF_W1 + k * VN_C1 + oc], xs[VN_S1 * t + k], a);
        f1[t][oc] = fmaxf(a, 0.0f);
    }
    __syncthreads();
    // conv2: output (t, oc) = relu(b2[oc] + sum_tap sum_ic w2[oc][ic][tap] f1[2 t - 1 + tap][ic]), tap-major
    for (int o = tid; o < VN_T2 * VN_C2; o += 256) {
        const int t = o / VN_C2, oc = o % VN_C2;
        float a = w[VN_OFF_B2 + oc];
        for (int tap = 0; tap < 3; tap++) {
            const int ti = 2 * t - 1 + tap;
            if (ti < 0 || ti >= VN_T1) continue;
            for (int ic = 0; ic < VN_C1; ic++) a = fmaf(w[VN_OFF_W2 + (tap * VN_C1 + ic) * VN_C2 + oc], f1[ti][ic], a);
        }
        f2[t][oc] = fmaxf(a, 0.0f);
    }
    __syncthreads();
    // two LSTM layers over the 6 steps (gate order i, f, g, o as torch.nn.LSTM); thread g owns gate row g
    for (int s = 0; s < VN_T2; s++) {
        for (int l = 0; l < 2; l++) {
            const float *wl = w + VN_OFF_L + l * VN_LSTM;
            const float *in = l == 0 ? f2[s] : hs[0];
            float a = wl[2 * VN_H * VN_G + tid];
            for (int k = 0; k < VN_H; k++) a = fmaf(wl[k * VN_G + tid], in[k], a);
            for (int k = 0; k < VN_H; k++) a = fmaf(wl[(VN_H + k) * VN_G + tid], hs[l][k], a);
            gs[tid] = a;
            __syncthreads();
            if (tid < VN_H) {
                const float ig = vn_sigmoid(gs[tid]), fg = vn_sigmoid(gs[VN_H + tid]), gg = tanhf(gs[2 * VN_H + tid]),
                            og = vn_sigmoid(gs[3 * VN_H + tid]);
                const float c = fg * cs[l][tid] + ig * gg;
                cs[l][tid] = c;
                hs[l][tid] = og * tanhf(c);
            }
            __syncthreads();
        }
        if (tid < 64) {                                   // y_s = wout . h1 + bout, summed over the lanes in a fixed tree
            float v = w[VN_OFF_WO + tid] * hs[1][tid];
            v = wave_sum(v);
            if (tid == 0) ys[s] = vn_sigmoid(v + w[VN_OFF_BO]);
        }
        __syncthreads();
    }
    if (tid < 2 * VN_H) {
        const int l = tid / VN_H, j = tid % VN_H;
        h_out[((int64_t)l * n + b) * VN_H + j] = hs[l][j];
        c_out[((int64_t)l * n + b) * VN_H + j] = cs[l][j];
    }
    if (tid == 0) {
        float p = 0.0f;
        for (int s = 0; s < VN_T2; s++) p += ys[s];
        prob[b] = p * (1.0f / VN_T2);
    }
}

// dsp.hip (ifh_ingest_block_net): the window step of the block driver, state updated in place
void launch_vadnet_slots(const float *win, const int32_t *slot, int n, const float *weights, float *h, float *c, float *prob, hipStream_t st)
{
    hipLaunchKernelGGL(k_vadnet, dim3(n), dim3(256), 0, st, win, slot, n, weights, h, c, h, c, prob);
}

}  // namespace ifh

using namespace ifh;

extern "C" int ifh_vadnet_weight_floats(void) { return VN_FLOATS; }

extern "C" int ifh_vadnet_prob(const float *x, int n, const float *weights, const float *h_in, const float *c_in, float *h_out,
                               float *c_out, float *prob, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(x && weights && h_in && c_in && h_out && c_out && prob);
    hipLaunchKernelGGL(k_vadnet, dim3(n), dim3(256), 0, as_stream(stream), x, (const int32_t *)nullptr, n, weights, h_in, c_in, h_out, c_out, prob);
    IFH_LAUNCH_CHECK("vadnet_prob");
    return IFH_OK;
}
