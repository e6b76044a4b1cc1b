// PixelUnshuffle(2) / PixelShuffle(2) over NCHW (model/Restromer.py:196,211).
// Pure permutations, HBM-bound: each element is read once and written once.
//   shuffle  : out[b, c, 2y+i, 2x+j] = in[b, 4c+2i+j, y, x]
//   unshuffle: out[b, 4c+2i+j, y, x] = in[b, c, 2y+i, 2x+j]
// A work item handles 4 consecutive x of the low-resolution side for one
// (b, c, i, y): two float4 on the low-res side (j = 0, 1) against 8 consecutive
// floats of one high-resolution row, so both sides move in 16-byte pieces and
// consecutive lanes touch consecutive addresses on both sides.
#include "kernels.hpp"

namespace rst {

// lo: [B][4C][H][W] ; hi: [B][C][2H][2W].  SHUFFLE: lo -> hi, else hi -> lo.
template <bool SHUFFLE, bool VEC>
__global__ void __launch_bounds__(256)
pixel_shuffle2_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int wq = (W + 3) / 4;
    const int xq = (int)(idx % wq);
    int64_t t = idx / wq;
    const int y = (int)(t % H); t /= H;
    const int i = (int)(t & 1); t >>= 1;
    const int c = (int)(t % C);
    const int64_t b = t / C;
    const int x0 = xq * 4;
    const int64_t lo0 = (((b * 4 * C + 4 * c + 2 * i + 0) * H + y) * (int64_t)W) + x0;
    const int64_t lo1 = lo0 + (int64_t)H * W;
    const int64_t hi = ((b * C + c) * 2 * H + 2 * y + i) * (int64_t)(2 * W) + 2 * x0;
    if constexpr (VEC) {
        if constexpr (SHUFFLE) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + lo0);
            const f32x4 bq = *reinterpret_cast<const f32x4*>(src + lo1);
            *reinterpret_cast<f32x4*>(dst + hi) = f32x4{a.x, bq.x, a.y, bq.y};
            *reinterpret_cast<f32x4*>(dst + hi + 4) = f32x4{a.z, bq.z, a.w, bq.w};
        } else {
            const f32x4 p = *reinterpret_cast<const f32x4*>(src + hi);
            const f32x4 q = *reinterpret_cast<const f32x4*>(src + hi + 4);
            *reinterpret_cast<f32x4*>(dst + lo0) = f32x4{p.x, p.z, q.x, q.z};
            *reinterpret_cast<f32x4*>(dst + lo1) = f32x4{p.y, p.w, q.y, q.w};
        }
    } else {
        for (int k = 0; k < 4 && x0 + k < W; ++k) {
            if constexpr (SHUFFLE) {
                dst[hi + 2 * k] = src[lo0 + k];
                dst[hi + 2 * k + 1] = src[lo1 + k];
            } else {
                dst[lo0 + k] = src[hi + 2 * k];
                dst[lo1 + k] = src[hi + 2 * k + 1];
            }
        }
    }
}

// C, H, W describe the high-channel (low-resolution) side: lo is [B][4C][H][W].
int pixel_shuffle2_launch(bool shuffle, const float* src, float* dst, int B, int C, int H, int W, hipStream_t stream) {
    RST_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "pixel_shuffle: empty problem");
    RST_REQUIRE(src && dst, "pixel_shuffle: null operand");
    const int64_t total = (int64_t)B * C * 2 * H * cdiv(W, 4);
    const int64_t grid = cdiv(total, 256);
    RST_REQUIRE(grid < (1ll << 31), "pixel_shuffle: grid too large");
    const bool vec = (W % 4 == 0) && aligned16(src) && aligned16(dst);
    dim3 gd((unsigned)grid), blk(256);
    ProfScope prof(stream, shuffle ? "pixel_shuffle2" : "pixel_unshuffle2", 0.0, 2.0 * 4 * 4 * B * C * H * W);
    if (shuffle) {
        if (vec) hipLaunchKernelGGL((pixel_shuffle2_kernel<true, true>), gd, blk, 0, stream, src, dst, C, H, W, total);
        else     hipLaunchKernelGGL((pixel_shuffle2_kernel<true, false>), gd, blk, 0, stream, src, dst, C, H, W, total);
    } else {
        if (vec) hipLaunchKernelGGL((pixel_shuffle2_kernel<false, true>), gd, blk, 0, stream, src, dst, C, H, W, total);
        else     hipLaunchKernelGGL((pixel_shuffle2_kernel<false, false>), gd, blk, 0, stream, src, dst, C, H, W, total);
    }
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst
