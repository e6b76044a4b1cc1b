// Device helpers shared by the channel-packed bf16 kernels (bf16_pk.hip, bf16_bwd.hip).
#pragma once
#include "kernels.hpp"

namespace rst {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {     // round to nearest even
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void unpack8(const u32x4 w, float* v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf_lo(w[i]); v[2 * i + 1] = bf_hi(w[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float* v) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16(v[2 * i], v[2 * i + 1]);
    return w;
}
__device__ __forceinline__ uint16_t to_bf16_bits(float x) { return (uint16_t)(pack_bf16(x, 0.f) & 0xffffu); }

// Position p of a packed buffer -> channel of the source tensor, or -1 for padding.  The source channels come in
// segments of `seg` channels, each padded to `seg_pad` positions (the two halves of the gated feed-forward).
// seg_pad < 0: two segments interleaved in groups of four -- block j of 8 positions holds channels 4j..4j+3 of the
// first segment and then the same channels of the second (the gated feed-forward: a lane that owns one block has
// both factors of gelu(x1) * x2).
__device__ __forceinline__ int src_channel(int p, int seg, int seg_pad, int total) {
    if (seg_pad < 0) {
        const int i = 4 * (p >> 3) + (p & 3), half = (p >> 2) & 1;
        return (i < seg && half * seg + i < total) ? half * seg + i : -1;
    }
    const int q = p / seg_pad, r = p % seg_pad;
    const int ch = q * seg + r;
    return (r < seg && ch < total) ? ch : -1;
}

}  // namespace rst
