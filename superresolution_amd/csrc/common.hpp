// Shared host/device helpers for the Restormer block kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>

#include "../../include/restormer_hip.h"

namespace rst {

// ---- error reporting --------------------------------------------------------
// Every C entry point returns an rst_status; the message of the last failure on
// the calling thread is kept for rst_last_error().
std::string& last_error_slot();

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

#define RST_REQUIRE(cond, ...)                                            \
    do {                                                                  \
        if (!(cond)) return ::rst::fail(RST_ERR_INVALID, __VA_ARGS__);    \
    } while (0)

#define RST_HIP(call)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return ::rst::fail(RST_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                               hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

#define RST_LAUNCH_CHECK()                                                                \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess)                                                             \
            return ::rst::fail(RST_ERR_HIP, "kernel launch failed: %s (%s:%d)",           \
                               hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

#define RST_TRY(expr)                 \
    do {                              \
        int s_ = (expr);              \
        if (s_ != RST_OK) return s_;  \
    } while (0)

// RST_XCD_WALK=0: kernels that re-read a neighbour's operand keep the plain workgroup order (for A/B timing)
inline int xcd_walk_enabled() {
    static const int on = !(getenv("RST_XCD_WALK") != nullptr && getenv("RST_XCD_WALK")[0] == '0');
    return on;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

// ---- per-(device, kernel) launch set-up, asked once -----------------------------
// CU count, the dynamic-LDS attribute and the occupancy of a kernel do not change between launches; the runtime
// calls behind them cost host time on every launch and are not allowed while a stream is being captured.
struct KernelSetup { int cus = 0, per_cu = 0; };
inline int kernel_setup(const void* kern, int threads, size_t lds, KernelSetup* out) {
    static std::mutex mu;
    static std::map<std::tuple<int, const void*, size_t>, KernelSetup> seen;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(RST_ERR_HIP, "hipGetDevice failed: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(mu);
    KernelSetup& slot = seen[std::make_tuple(dev, kern, lds)];
    if (slot.cus == 0) {
        int cus = 256, per_cu = 1;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess && lds > 64 * 1024)
            e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds);
        if (e != hipSuccess) return fail(RST_ERR_HIP, "kernel set-up failed: %s", hipGetErrorString(e));
        slot.cus = cus; slot.per_cu = per_cu < 1 ? 1 : per_cu;
    }
    *out = slot;
    return RST_OK;
}

// ---- optional per-kernel timing (prof.hip) -------------------------------------
bool prof_enabled();
class ProfScope {
public:
    ProfScope(hipStream_t s, const char* label, double flops, double bytes);
    ~ProfScope();
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
private:
    hipStream_t stream_; const char* label_; double flops_, bytes_; bool on_; void* e0_;
};

// ---- device helpers ---------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Workgroups are dealt to the 8 XCDs round-robin by their linear id (ids 8 apart share an XCD and its L2).  Maps the
// linear id `bid` of a grid of `n` workgroups to a logical id such that every XCD walks one contiguous range of
// logical ids in order: neighbouring logical ids then run on one XCD at about the same time, and what they both read
// comes from HBM once.  A bijection on [0, n) for any n; placement is a matter of speed only.
__device__ __forceinline__ int64_t xcd_walk(int64_t bid, int64_t n) {
    const int64_t q = n >> 3, r = n & 7;
    const int64_t x = bid & 7, local = bid >> 3;
    return x * q + (x < r ? x : r) + local;
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// d/dx [ 0.5 x (1 + erf(x/sqrt2)) ] = 0.5 (1 + erf(x/sqrt2)) + x * exp(-x^2/2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) +
           x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

}  // namespace rst
