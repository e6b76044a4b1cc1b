// Peak micro-benchmarks: what this GPU delivers on a streaming copy and on bare matrix-instruction loops.
// Measurement aids (no counterpart in the reference); tools/peaks.py times them with HIP events and bench.py quotes
// the result next to the paper peaks its roofline fractions are computed from.
#include "kernels.hpp"

namespace rst {

typedef _Float16 pk_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) peak_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}

__global__ void __launch_bounds__(256) peak_read_kernel(const f32x4* __restrict__ src, float* __restrict__ out, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) s += __builtin_nontemporal_load(&src[i]);
    const float t = s.x + s.y + s.z + s.w;
    if (t == 123.456f) out[0] = t;                      // keeps the loads alive, practically never taken
}

// KIND 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_16x16x32_bf16, 2: v_mfma_f32_16x16x32_f16; 8 independent accumulators,
// operands in registers with lane-dependent (non-trivial) values
template <int KIND>
__global__ void __launch_bounds__(256) peak_mfma_kernel(float* __restrict__ out, int iters, float seed) {
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float base = seed + 0.001f * (float)(threadIdx.x & 63) + 0.37f * (float)(blockIdx.x & 7);
    if constexpr (KIND == 0) {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = base + 0.11f * j; b[j] = 1.0f / (base + 1.0f + j); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[j], 0, 0, 0);
        }
    } else if constexpr (KIND == 1) {
        pk_bf16x8 a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[j][e] = (__bf16)(base + 0.11f * j + 0.013f * e); b[j][e] = (__bf16)(1.0f / (base + 1.0f + j + e)); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j >> 1) & 3], acc[j], 0, 0, 0);
        }
    } else {
        pk_f16x8 a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[j][e] = (_Float16)(base + 0.11f * j + 0.013f * e); b[j][e] = (_Float16)(1.0f / (base + 1.0f + j + e)); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j & 3], b[(j >> 1) & 3], acc[j], 0, 0, 0);
        }
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) s += acc[j];
    out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}

}  // namespace rst

using namespace rst;

extern "C" {

int rst_peak_copy(const void* src, void* dst, size_t bytes, int32_t read_only, void* stream) {
    RST_REQUIRE(src && dst && bytes >= 16 && aligned16(src) && aligned16(dst), "peak_copy: bad buffers");
    const int64_t n16 = (int64_t)(bytes / 16);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (read_only) hipLaunchKernelGGL(peak_read_kernel, dim3(256 * 16), dim3(256), 0, st, static_cast<const f32x4*>(src), static_cast<float*>(dst), n16);
    else hipLaunchKernelGGL(peak_copy_kernel, dim3(256 * 16), dim3(256), 0, st, static_cast<const f32x4*>(src), static_cast<f32x4*>(dst), n16);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

// out: blocks * 256 floats.  FLOPs of one call = blocks * 4 waves * iters * 8 * 2 * 16 * 16 * (4 or 32).
int rst_peak_mfma(int32_t kind, int32_t blocks, int32_t iters, float* out, void* stream) {
    RST_REQUIRE(out && blocks > 0 && iters > 0 && kind >= 0 && kind <= 2, "peak_mfma: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (kind == 0) hipLaunchKernelGGL(peak_mfma_kernel<0>, dim3(blocks), dim3(256), 0, st, out, iters, 0.5f);
    else if (kind == 1) hipLaunchKernelGGL(peak_mfma_kernel<1>, dim3(blocks), dim3(256), 0, st, out, iters, 0.5f);
    else hipLaunchKernelGGL(peak_mfma_kernel<2>, dim3(blocks), dim3(256), 0, st, out, iters, 0.5f);
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // extern "C"
