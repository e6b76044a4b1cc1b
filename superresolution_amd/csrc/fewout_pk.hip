// Dense 3x3 convolution with at most 4 output channels on a channel-packed bf16 input ([B][Cb][H*W][8], bf16_pk.hip)
// with an fp32 NCHW result: the last layer of the network on the bf16 storage path (model/Restromer.py:381-383,
// 420-431: `self.output(...)` plus the bilinear x2 of the network's input, or a plain residual).  fp32 weights, fp32
// sums, and the image that leaves the network never passes through a bf16 container.
//
// One thread = 2 neighbouring pixel columns x a strip of ROWS rows x ALL output channels; the channel blocks of the
// input are walked one after the other and their contributions add up in the thread's COUT x ROWS x 2 running sums.
// Per channel block an input row is four 16-byte chunks (x-1 .. x+2; the next row's already in flight); every row adds
// to the three output rows it touches.  The 8 x 9 x COUT weights of a channel block are wave-uniform (scalar loads).
// Branch-free loads: chunks outside the image read a valid address and are zeroed where they are used.
// grid = (ceil(W/2 * strips / 256), 1, B)
#include "kernels.hpp"
#include "bf16_common.hpp"

namespace rst {

namespace {

constexpr int FO_ROWS = 4;

template <int COUT>
__global__ void __launch_bounds__(256)
fewout_pk_kernel(const bf16_t* __restrict__ x, int64_t x_bs, const float* __restrict__ w, const float* __restrict__ bias,
                 const float* __restrict__ res, const float* __restrict__ up_src, float* __restrict__ y,
                 int Cin, int Cb, int H, int W) {
    const int b = blockIdx.z;
    const int W2 = W / 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int strip = t / W2, x0 = (t % W2) * 2;
    const int y0 = strip * FO_ROWS;
    if (y0 >= H) return;
    const int64_t N = (int64_t)H * W;
    const u32x4* X = reinterpret_cast<const u32x4*>(x + (int64_t)b * x_bs);
    const bool hasl = x0 > 0, hasr = x0 + 2 < W;
    const int xl = hasl ? x0 - 1 : x0, xr = hasr ? x0 + 2 : x0 + 1;
    const u32x4 zero = {0u, 0u, 0u, 0u};

    float o[COUT][FO_ROWS][2];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < FO_ROWS; ++r) { o[co][r][0] = bv; o[co][r][1] = bv; }
    }
    auto fetch = [&](u32x4 (&dst)[4], const u32x4* plane, int i) {
        const u32x4* row = plane + (int64_t)(i < 0 ? 0 : (i < H ? i : H - 1)) * W;
        dst[0] = row[xl]; dst[1] = row[x0]; dst[2] = row[x0 + 1]; dst[3] = row[xr];
    };
    for (int cb = 0; cb < Cb; ++cb) {
        const u32x4* plane = X + (int64_t)cb * N;
        u32x4 cur[4], nxt[4];
        fetch(cur, plane, y0 - 1);
#pragma unroll
        for (int ii = 0; ii < FO_ROWS + 2; ++ii) {
            const int i = y0 - 1 + ii;
            if (ii + 1 < FO_ROWS + 2) fetch(nxt, plane, i + 1);
            const bool rv = i >= 0 && i < H;
            float c[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool ok = rv && (k > 0 || hasl) && (k < 3 || hasr);
                unpack8(ok ? cur[k] : zero, c[k]);
            }
            // input row ii (image row y0 - 1 + ii) is window row dy of output row ii - dy
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = ii - dy;
                if (r < 0 || r >= FO_ROWS) continue;
#pragma unroll
                for (int co = 0; co < COUT; ++co)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int ci = 8 * cb + j;
                        // (input channels past Cin -- the padding of the last channel block -- meet zero weights)
                        const float* wk = w + ((int64_t)co * Cin + (ci < Cin ? ci : 0)) * 9 + dy * 3;
                        const float live = ci < Cin ? 1.f : 0.f;      // (a multiplication: a select here turns the scalar weight loads into masked vector loads, 5x slower)
                        const float k0 = wk[0] * live, k1 = wk[1] * live, k2 = wk[2] * live;
                        o[co][r][0] = fmaf(k0, c[0][j], fmaf(k1, c[1][j], fmaf(k2, c[2][j], o[co][r][0])));
                        o[co][r][1] = fmaf(k0, c[1][j], fmaf(k1, c[2][j], fmaf(k2, c[3][j], o[co][r][1])));
                    }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
        }
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const int64_t off = ((int64_t)b * COUT + co) * N;
#pragma unroll
        for (int r = 0; r < FO_ROWS; ++r) {
            const int yy = y0 + r;
            if (yy >= H) continue;
            f32x2 v = {o[co][r][0], o[co][r][1]};
            if (res != nullptr) {
                const f32x2 rv = *reinterpret_cast<const f32x2*>(res + off + (int64_t)yy * W + x0);
                v += rv;
            }
            if (up_src != nullptr) {
                // + bilinear x2 of the low-resolution tensor (align_corners = False): source coordinate o / 2 - 1/4
                // clamped at 0, weights (1/4, 3/4) or (3/4, 1/4), the far index clamped at the border
                const int h2 = H / 2, w2 = W / 2;
                const float* lp = up_src + ((int64_t)b * COUT + co) * h2 * w2;
                const float sy = fmaxf(0.5f * (float)yy - 0.25f, 0.f);
                const int iy0 = (int)sy, iy1 = iy0 + 1 < h2 ? iy0 + 1 : h2 - 1;
                const float ly1 = sy - (float)iy0, ly0 = 1.0f - ly1;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float sx = fmaxf(0.5f * (float)(x0 + j) - 0.25f, 0.f);
                    const int ix0 = (int)sx, ix1 = ix0 + 1 < w2 ? ix0 + 1 : w2 - 1;
                    const float lx1 = sx - (float)ix0, lx0 = 1.0f - lx1;
                    const float p00 = lp[(int64_t)iy0 * w2 + ix0], p01 = lp[(int64_t)iy0 * w2 + ix1];
                    const float p10 = lp[(int64_t)iy1 * w2 + ix0], p11 = lp[(int64_t)iy1 * w2 + ix1];
                    v[j] += ly0 * (lx0 * p00 + lx1 * p01) + ly1 * (lx0 * p10 + lx1 * p11);
                }
            }
            *reinterpret_cast<f32x2*>(y + off + (int64_t)yy * W + x0) = v;
        }
    }
}

}  // namespace

int fewout_pk_launch(const bf16_t* x, const float* w, const float* bias, const float* res, const float* up_src, float* y,
                     int B, int Cin, int Cout, int H, int W, hipStream_t stream) {
    RST_REQUIRE(x && w && y, "bf16 conv3x3 (few outputs): null operand");
    RST_REQUIRE(Cout >= 1 && Cout <= 4, "bf16 conv3x3 (few outputs): %d output channels (1 to 4)", Cout);
    RST_REQUIRE(W % 2 == 0 && W >= 2 && H >= 1 && B >= 1 && B <= 65535 && Cin >= 1, "bf16 conv3x3 (few outputs): bad shape %dx%dx%dx%d", B, Cin, H, W);
    RST_REQUIRE(!(res && up_src), "bf16 conv3x3 (few outputs): one residual form at a time");
    RST_REQUIRE(up_src == nullptr || H % 2 == 0, "bf16 conv3x3 (few outputs): the bilinear residual needs an even height");
    RST_REQUIRE(aligned16(x) && (reinterpret_cast<uintptr_t>(y) & 7u) == 0 && (res == nullptr || (reinterpret_cast<uintptr_t>(res) & 7u) == 0),
                "bf16 conv3x3 (few outputs): misaligned operand");
    const int Cb = (int)cdiv(Cin, 8);
    const int64_t px = (int64_t)B * H * W;
    const dim3 grid((unsigned)cdiv((int64_t)(W / 2) * cdiv(H, FO_ROWS), 256), 1, (unsigned)B);
    ProfScope prof(stream, "pk_conv3x3_fewout", 18.0 * Cin * Cout * px, px * (2.0 * Cb * 8 + 4.0 * Cout * (res ? 2.0 : 1.0)));
    const int64_t x_bs = (int64_t)Cb * H * W * 8;
#define FO_GO(n) hipLaunchKernelGGL((fewout_pk_kernel<n>), grid, dim3(256), 0, stream, x, x_bs, w, bias, res, up_src, y, Cin, Cb, H, W)
    switch (Cout) { case 1: FO_GO(1); break; case 2: FO_GO(2); break; case 3: FO_GO(3); break; default: FO_GO(4); break; }
#undef FO_GO
    RST_LAUNCH_CHECK();
    return RST_OK;
}

}  // namespace rst

extern "C" int32_t rst_bf16_conv3x3_fewout_fwd(const void* x_packed, const float* weight, const float* bias, const float* residual,
                                               const float* lowres, float* y, int32_t B, int32_t Cin, int32_t Cout, int32_t H,
                                               int32_t W, void* stream) {
    return rst::fewout_pk_launch(static_cast<const rst::bf16_t*>(x_packed), weight, bias, residual, lowres, y, B, Cin, Cout, H, W,
                                 static_cast<hipStream_t>(stream));
}
