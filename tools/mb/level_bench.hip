// Standalone timing of ifh_resblock_level_bf16 (csrc/level.hip + csrc/level_pipe.hip), no Python: random bf16 rows and weights,
// the C = 32 level of the vocoder (T = 3072 rows per chunk).  Built by tools/r06_level_variants.sh in several compile variants.
//     level_bench [nchunks] [taps: 3 | 7 | 11 | 0 = all three blocks] [reps]        IFH_LEVEL_BARRIER=1: the barrier form
//     IFH_LEVEL_ABL=16 (builds with -DLV_DEV_ABL): phase clocks of wave 0 per convolution
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../infernos_amd/csrc/common.h"

namespace ifh {
std::atomic<int> g_cu_budget{0};
void set_error(const std::string &msg) { fprintf(stderr, "ifh error: %s\n", msg.c_str()); }
int fail(int code, const std::string &msg) { set_error(msg); return code; }
int check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return IFH_OK;
    fprintf(stderr, "hip error at %s: %s\n", what, hipGetErrorString(e));
    return IFH_EHIP;
}
}  // namespace ifh

static uint16_t f2bf(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}
static uint64_t g_lcg = 88172645463325252ull;       // own generator: the HIP runtime's start-up may call rand()
static float frand()
{
    g_lcg = g_lcg * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((g_lcg >> 40) & 0xffffff) / 8388608.0f - 1.f;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1280, taps = argc > 2 ? atoi(argv[2]) : 0, reps = argc > 3 ? atoi(argv[3]) : 10;
    const int T = 3072, C = 32;
    const size_t elems = (size_t)n * T * C;
    std::vector<uint16_t> hx(elems);
    for (size_t i = 0; i < elems; i++) hx[i] = f2bf(frand() * 1.5f);
    uint16_t *x, *out;
    CK(hipMalloc(&x, elems * 2));
    CK(hipMalloc(&out, elems * 2));
    CK(hipMemcpy(x, hx.data(), elems * 2, hipMemcpyHostToDevice));
    CK(hipMemset(out, 0, elems * 2));
    ifh_level_desc d;
    memset(&d, 0, sizeof(d));
    const int ks[3] = {3, 7, 11};
    d.nblocks = taps ? 1 : 3;
    double flops = 0;
    for (int j = 0; j < d.nblocks; j++) {
        const int k = taps ? taps : ks[j];
        d.taps[j] = k;
        const size_t wn = (size_t)6 * k * 2 * 512;                       // bf16 elements: 6 convolutions x k k-steps x 2 fragments x 1 KB
        std::vector<uint16_t> hw(wn);
        const float sc = 1.0f / sqrtf((float)(C * k));
        for (size_t i = 0; i < wn; i++) hw[i] = f2bf(frand() * 1.7f * sc);
        std::vector<float> hb(6 * C);
        for (auto &v : hb) v = frand() * 0.1f;
        void *w;
        float *b;
        CK(hipMalloc(&w, wn * 2));
        CK(hipMalloc(&b, hb.size() * 4));
        CK(hipMemcpy(w, hw.data(), wn * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        d.wstream[j] = w;
        d.bias[j] = b;
        flops += 6 * 2.0 * n * T * C * C * k;
    }
    d.x = x; d.x_bstride = (int64_t)T * C; d.out = out; d.out_bstride = (int64_t)T * C;
    d.c = C; d.t = T; d.nbatch = n; d.slope = 0.1f; d.out_scale = 1.0f / 3.0f; d.accumulate = 0;
    unsigned long long *prof = nullptr;
    const bool abl = getenv("IFH_LEVEL_ABL") && atoi(getenv("IFH_LEVEL_ABL")) == 16;
    if (abl) {
        CK(hipMalloc(&prof, 16 * 8));
        CK(hipMemset(prof, 0, 16 * 8));
        d.debug_prof = prof;
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int i = 0; i < 3; i++)
        if (ifh_resblock_level_bf16(&d, (ifh_stream_t)st) != IFH_OK) return 2;
    CK(hipStreamSynchronize(st));
    if (abl) CK(hipMemset(prof, 0, 16 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++)
        if (ifh_resblock_level_bf16(&d, (ifh_stream_t)st) != IFH_OK) return 2;
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double t = ms * 1e-3 / reps;
    // determinism: the same launch into a second buffer, element-wise comparison
    {
        uint16_t *out2;
        CK(hipMalloc(&out2, elems * 2));
        CK(hipMemset(out2, 0, elems * 2));
        ifh_level_desc d2 = d;
        d2.out = out2;
        if (ifh_resblock_level_bf16(&d2, (ifh_stream_t)st) != IFH_OK) return 2;
        CK(hipStreamSynchronize(st));
        std::vector<uint16_t> h1(elems), h2(elems);
        CK(hipMemcpy(h1.data(), out, elems * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2.data(), out2, elems * 2, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < elems; i++)
            if (h1[i] != h2[i]) { if (!bad) first = i; bad++; }
        printf("   two launches differ in %zu of %zu elements", bad, elems);
        if (bad) printf(" (first: chunk %zu row %zu channel %zu: %04x vs %04x)", first / (T * C), first / C % T, first % C, h1[first], h2[first]);
        printf("\n");
        size_t nan = 0;
        for (size_t i = 0; i < elems; i++) nan += (h1[i] & 0x7f80) == 0x7f80;
        printf("   inf/nan elements: %zu\n", nan);
    }
    // a checksum of the output so that variants can be compared for equal results
    std::vector<uint16_t> ho(elems);
    CK(hipMemcpy(ho.data(), out, elems * 2, hipMemcpyDeviceToHost));
    unsigned long long cs = 1469598103934665603ull;
    for (size_t i = 0; i < elems; i++) cs = (cs ^ ho[i]) * 1099511628211ull;
    printf("%s n=%d taps=%d: %8.1f us  %7.1f TF/s  checksum %016llx\n", getenv("IFH_LEVEL_BARRIER") ? "barrier  " : "pipelined", n, taps, t * 1e6,
           flops / t / 1e12, cs);
    if (abl) {
        unsigned long long pr[16];
        CK(hipMemcpy(pr, prof, sizeof(pr), hipMemcpyDeviceToHost));
        const double nc = pr[6] ? (double)pr[6] : 1.0;
        printf("   per convolution (wave 0 of every workgroup, shader clocks): [0] %.0f  [1] %.0f  [2] %.0f  [3] %.0f  [4] %.0f  [5] %.0f  | block top %.0f per block; %.0f convolutions\n",
               pr[0] / nc, pr[1] / nc, pr[2] / nc, pr[3] / nc, pr[4] / nc, pr[5] / nc, pr[7] * 6 / nc, nc);
    }
    return 0;
}
