// Sensor sample -> training pair on the GPU: the reference's dataset item (data/loader.py:33-52) for a batch that
// is already in device memory.  raw is B x H x W x C (interleaved channels, C <= 4; uint16 counts or float32),
// max_val the white level per sample.  hr = raw / max_val as B x C x H2 x W2 (an odd trailing row / column is
// dropped, data/loader.py:45-48), lr = its 2 x 2 average pool (utils/util.py:313-321), both planar fp32.
// The reference divides and pools in float64 (numpy promotes float32 / 0-d int64 to float64) and casts to
// float32 afterwards (model/train.py:76-77): so does this kernel, which keeps the result bit-identical.
#include "kernels.hpp"

namespace rst {

template <typename T, int C>
__global__ void __launch_bounds__(256)
raw_prepare_kernel(const T* __restrict__ raw, const float* __restrict__ max_val, float* __restrict__ hr, float* __restrict__ lr,
                   int H, int W, int H2, int W2) {
    const int lw = W2 / 2, lh = H2 / 2;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, b = blockIdx.z;
    if (x >= lw) return;
    const double inv_den = (double)max_val[b];
    const T* src = raw + (((int64_t)b * H + 2 * y) * W + 2 * x) * C;
    double v[2][2][C];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c = 0; c < C; ++c) v[r][p][c] = (double)(float)src[((int64_t)r * W + p) * C + c] / inv_den;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float* h0 = hr + (((int64_t)b * C + c) * H2 + 2 * y) * W2 + 2 * x;
        h0[0] = (float)v[0][0][c]; h0[1] = (float)v[0][1][c];
        h0[W2] = (float)v[1][0][c]; h0[W2 + 1] = (float)v[1][1][c];
        // pooled in float64 in the order of a row-major window walk, then rounded once
        const double s = ((v[0][0][c] + v[0][1][c]) + v[1][0][c]) + v[1][1][c];
        lr[(((int64_t)b * C + c) * lh + y) * lw + x] = (float)(s / 4.0);
    }
}

int raw_prepare_launch(const void* raw, int raw_is_u16, const float* max_val, float* hr, float* lr,
                       int B, int H, int W, int C, hipStream_t stream) {
    RST_REQUIRE(raw && max_val && hr && lr, "raw_prepare: null operand");
    RST_REQUIRE(B > 0 && H >= 2 && W >= 2 && C >= 1 && C <= 4, "raw_prepare: shape B=%d H=%d W=%d C=%d (C <= 4)", B, H, W, C);
    RST_REQUIRE(B <= 65535 && H / 2 <= 65535, "raw_prepare: grid out of range");
    const int H2 = H & ~1, W2 = W & ~1;
    dim3 blk(256), grid((unsigned)cdiv(W2 / 2, 256), (unsigned)(H2 / 2), (unsigned)B);
    const double px = (double)B * H2 * W2;
    ProfScope prof(stream, "raw_prepare", 3.0 * px * C, px * C * ((raw_is_u16 ? 2.0 : 4.0) + 4.0 + 1.0));
#define RP(T, CC) hipLaunchKernelGGL((raw_prepare_kernel<T, CC>), grid, blk, 0, stream, static_cast<const T*>(raw), max_val, hr, lr, H, W, H2, W2)
    if (raw_is_u16) { if (C == 4) RP(uint16_t, 4); else if (C == 3) RP(uint16_t, 3); else if (C == 2) RP(uint16_t, 2); else RP(uint16_t, 1); }
    else            { if (C == 4) RP(float, 4); else if (C == 3) RP(float, 3); else if (C == 2) RP(float, 2); else RP(float, 1); }
#undef RP
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst

extern "C" int rst_raw_prepare(const void* raw, int32_t raw_is_u16, const float* max_val, float* hr, float* lr,
                               int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    return rst::raw_prepare_launch(raw, raw_is_u16, max_val, hr, lr, B, H, W, C, static_cast<hipStream_t>(stream));
}
