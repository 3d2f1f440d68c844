// common.h -- shared helpers for the gfx950 kernels of libinfernos_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>
#include <string>

#include "../../include/infernos_hip.h"

namespace ifh {

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);
int check_hip(hipError_t e, const char *what);

#define IFH_CHECK_ARG(cond)                                                        \
    do {                                                                           \
        if (!(cond)) return ::ifh::fail(IFH_EINVAL, std::string(__func__) + ": " #cond); \
    } while (0)

#define IFH_LAUNCH_CHECK(what)                                                     \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) return ::ifh::check_hip(e__, what);                 \
    } while (0)

static inline hipStream_t as_stream(ifh_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute acts on the CURRENT device: a process that drives several GPUs (one worker thread per device behind the
// actors) must set a kernel's dynamic-LDS limit once per device, not once per process.  One DeviceOnce per call site (a
// function-local static); worker threads of different devices may pass through it concurrently, hence the atomic mask.  A
// device whose index cannot be read, or above 63, is never marked: its attribute is simply set again on every call.
struct DeviceOnce {
    std::atomic<unsigned long long> mask{0};
    bool needed(int *dev_out)
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = -1;
        *dev_out = dev;
        return dev < 0 || !((mask.load(std::memory_order_acquire) >> dev) & 1ull);
    }
    void done(int dev)
    {
        if (dev >= 0) mask.fetch_or(1ull << dev, std::memory_order_release);
    }
};

// ifh_set_cu_budget (misc.hip): CUs the persistent kernels may count on (0 = all of the device's)
extern std::atomic<int> g_cu_budget;

// compute units of the CURRENT device (cached per device index), capped by the CU budget
static inline int device_cu_count_physical();
static inline int device_cu_count()
{
    const int n = device_cu_count_physical(), b = g_cu_budget.load(std::memory_order_relaxed);
    return (b > 0 && b < n) ? b : n;
}
static inline int device_cu_count_physical()
{
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64) cache[dev].store(prop.multiProcessorCount, std::memory_order_relaxed);
    return prop.multiProcessorCount;
}

// vadnet.hip: one recurrent-VAD window step for n calls (window rows by slot, state rows [2][n][64] by call index, updated in place)
void launch_vadnet_slots(const float *win, const int32_t *slot, int n, const float *weights, float *h, float *c, float *prob, hipStream_t st);

constexpr int kWave = 64;

// ---- bf16 helpers (raw uint16 storage) ------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f)
{
    // native conversion (v_cvt_pk_bf16_f32 on gfx950): round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// ---- wave / block reductions ------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// order-preserving float <-> int map for atomicMax on floats
__device__ __forceinline__ int float_to_ordered(float f)
{
    int i = __float_as_int(f);
    return (i >= 0) ? i : (i ^ 0x7fffffff);
}
__device__ __forceinline__ float ordered_to_float(int i)
{
    return __int_as_float((i >= 0) ? i : (i ^ 0x7fffffff));
}

}  // namespace ifh
